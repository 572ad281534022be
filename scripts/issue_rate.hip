// Clocks per wave64 integer VALU instruction on gfx950, measured (round-2 review item: valu_issue_frac hung on an
// assumed constant).  Build: hipcc --offload-arch=gfx950 -O2 -o scripts/issue_rate scripts/issue_rate.hip
// Run on the GPU box: scripts/issue_rate > gpurun_out/issue_rate.json
//
// For each instruction kind a kernel runs ITER x 64 instances of it back to back on eight independent register
// chains (so latency never binds) and brackets the loop with s_memtime (tick = shader cycle, MI355X_MICROARCH.md
// "Per-instruction cycle constants").  Grids: one block of 64 * W threads on ONE CU for W = 1 .. 16 waves (waves
// go to the CU's 4 SIMDs cyclically: W = 4 is one wave per SIMD, W = 8 two, W = 16 four); the per-SIMD issue
// cost is  (max over waves of the bracket) / (instructions one SIMD issued) = cycles / (ITER * 64 * W / min(W,4)...)
// -- reported per wave (what one wave sees) and per SIMD (what the pipe sustains).  A whole-chip launch (2048
// blocks of 256) timed with hipEvents gives the same figure against wall time, i.e. at the clock the chip
// actually sustains, which is what a kernel's SQ_ACTIVE_INST_VALU x k / (GRBM_GUI_ACTIVE x SIMDs) needs.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 512;    // loop trips
constexpr int UNROLL = 64;   // instructions per trip (8 chains x 8)

enum Kind { ADD = 0, CNDMASK, BFE, CMP_CND, MINMAX, LSHL_ADD, AND_OR, ADD3, MAD24, MOV, XOR_DEP, SUB, AND, OR, XOR, LSHLREV, LSHRREV, ASHRREV, MIN, MAXI, MED3, MIN3, CMP, CMP64, CMP64_CND, CND_S, ADD_S, ADD_LIT, ADDCO, ADDC, MUL_LO, MUL24, PERM, ALIGNBIT, BFI, LSHL_OR, OR3, ADD_LSHL, FMA, MOV_DPP, READFL, MBCNT, CVT, N_KINDS };
static const char *kind_name[N_KINDS] = {"v_add_u32", "v_cndmask_b32 (vcc not written in the loop)", "v_bfe_u32", "v_cmp_lt_i32 vcc + v_cndmask_b32 vcc", "v_min_u32 + v_max_u32 (dependent pair)", "v_lshl_add_u32", "v_and_or_b32", "v_add3_u32", "v_mad_u32_u24", "v_mov_b32", "v_xor_b32 (ONE dependent chain)", "v_sub_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_min_u32", "v_max_i32", "v_med3_i32", "v_min3_u32", "v_cmp_lt_i32 vcc (alone)", "v_cmp_lt_i32 s[20:21] (e64, alone)", "v_cmp_lt_i32 s[20:21] + v_cndmask_b32 s[20:21]", "v_cndmask_b32 s[20:21] (not written in the loop)", "v_add_u32 with an SGPR source", "v_add_u32 with a 32-bit literal", "v_add_co_u32 vcc", "v_add_co_u32 + v_addc_co_u32 (64-bit add)", "v_mul_lo_u32", "v_mul_u32_u24", "v_perm_b32", "v_alignbit_b32", "v_bfi_b32", "v_lshl_or_b32", "v_or3_b32", "v_add_lshl_u32", "v_fma_f32", "v_mov_b32_dpp row_shr:1", "v_readfirstlane_b32 (to s22)", "v_mbcnt_lo_u32_b32", "v_cvt_f32_u32"};
static const int insts_per_slot[N_KINDS] = {1, 1, 1, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int K> __device__ __forceinline__ void body(uint32_t (&r)[8], uint32_t a, uint32_t b) {
  const uint32_t sg = __builtin_amdgcn_readfirstlane(b);
#define ONE(i) \
  if (K == ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == BFE) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == CMP_CND) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == MINMAX) asm volatile("v_min_u32 %0, %0, %1\n\tv_max_u32 %0, %0, %2" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == MAD24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == XOR_DEP) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[0]) : "v"(a)); \
  else if (K == SUB) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == OR) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == LSHLREV) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == LSHRREV) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == ASHRREV) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == MIN) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == MAXI) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == MED3) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == MIN3) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == CMP) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == CMP64) asm volatile("v_cmp_lt_i32 s[20:21], %0, %1" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == CMP64_CND) asm volatile("v_cmp_lt_i32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %2, s[20:21]" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == CND_S) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == ADD_S) asm volatile("v_add_u32 %0, %3, %0" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == ADD_LIT) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == ADDCO) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == ADDC) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == MUL24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == OR3) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == ADD_LSHL) asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == READFL) asm volatile("v_readfirstlane_b32 s22, %0" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == MBCNT) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  else if (K == CVT) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r[i]) : "v"(a), "v"(b), "s"(sg) : "vcc", "s20", "s21", "s22"); \
  ;
  R8(ONE) R8(ONE) R8(ONE) R8(ONE) R8(ONE) R8(ONE) R8(ONE) R8(ONE)
#undef ONE
}

template <int K> __global__ void rate_kernel(uint32_t *out, unsigned long long *cycles, uint32_t a, uint32_t b) {
  uint32_t r[8];
  for (int i = 0; i < 8; i++) r[i] = threadIdx.x * 8u + (uint32_t)i + a;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();  // s_memtime
  for (int it = 0; it < ITER; it++) body<K>(r, a, b);
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int K> void run_kind(std::string &json, uint32_t *d_out, unsigned long long *d_cyc, double clock_mhz) {
  char buf[512];
  json += std::string("  {\"inst\": \"") + kind_name[K] + "\", \"insts_per_slot\": " + std::to_string(insts_per_slot[K]) + ", \"one_cu\": [";
  const double n_inst = (double)ITER * UNROLL * insts_per_slot[K];
  bool first = true;
  for (int W : {1, 2, 4, 8, 16}) {
    rate_kernel<K><<<1, 64 * W>>>(d_out, d_cyc, 3u, 5u);  // warm
    rate_kernel<K><<<1, 64 * W>>>(d_out, d_cyc, 3u, 5u);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(W);
    CHECK(hipMemcpy(h.data(), d_cyc, W * 8, hipMemcpyDeviceToHost));
    unsigned long long mx = 0, mn = ~0ull;
    for (auto c : h) { mx = c > mx ? c : mx; mn = c < mn ? c : mn; }
    const int per_simd = (W + 3) / 4;
    snprintf(buf, sizeof buf, "%s{\"waves\": %d, \"waves_per_simd\": %d, \"cycles_max\": %llu, \"cycles_min\": %llu, "
             "\"cycles_per_inst_per_wave\": %.3f, \"cycles_per_inst_per_simd\": %.3f}",
             first ? "" : ", ", W, per_simd, mx, mn, (double)mx / n_inst, (double)mx / (n_inst * per_simd));
    json += buf;
    first = false;
  }
  json += "], \"whole_chip\": [";
  first = true;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: blocks of 256 threads, wps blocks per CU
    const int blocks = 256 * wps;
    rate_kernel<K><<<blocks, 256>>>(d_out, d_cyc, 3u, 5u);
    CHECK(hipDeviceSynchronize());
    const int reps = 20;
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) rate_kernel<K><<<blocks, 256>>>(d_out, d_cyc, 3u, 5u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)blocks * 4);
    CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto c : h) avg += (double)c;
    avg /= (double)h.size();
    // wave-instructions one SIMD issued per launch = n_inst * wps; wall clocks per launch at clock_mhz
    const double wall_cycles = (double)ms / reps * 1e-3 * clock_mhz * 1e6;
    snprintf(buf, sizeof buf, "%s{\"waves_per_simd\": %d, \"ms_per_launch\": %.4f, \"s_memtime_cycles_avg\": %.0f, "
             "\"s_memtime_cycles_per_inst_per_simd\": %.3f, \"wall_cycles_per_inst_per_simd_at_max_clock\": %.3f, "
             "\"wave_insts_per_s_per_simd\": %.4g}",
             first ? "" : ", ", wps, ms / reps, avg, avg / (n_inst * wps), wall_cycles / (n_inst * wps),
             n_inst * wps / ((double)ms / reps * 1e-3));
    json += buf;
    first = false;
  }
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
  json += "]}";
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  uint32_t *d_out;
  unsigned long long *d_cyc;
  CHECK(hipMalloc(&d_out, 2048 * 1024 * 4));
  CHECK(hipMalloc(&d_cyc, 2048 * 16 * 8));
  const double clock_mhz = prop.clockRate / 1000.0;
  std::string json = "{\n \"device\": \"" + std::string(prop.gcnArchName) + "\", \"cus\": " + std::to_string(prop.multiProcessorCount) +
                     ", \"max_clock_mhz\": " + std::to_string(clock_mhz) + ", \"iter\": " + std::to_string(ITER) + ", \"unroll\": " +
                     std::to_string(UNROLL) + ",\n \"note\": \"cycles = s_memtime ticks (shader clock) around ITER x UNROLL instances; per_simd = per "
                     "instruction one SIMD issued (waves share the pipe); whole_chip wall figure assumes max clock, the s_memtime one does not\",\n \"results\": [\n";
  run_kind<ADD>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<CNDMASK>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<BFE>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<CMP_CND>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<MINMAX>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<LSHL_ADD>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<AND_OR>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<ADD3>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<MAD24>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<MOV>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<XOR_DEP>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<SUB>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<AND>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<OR>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<XOR>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<LSHLREV>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<LSHRREV>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<ASHRREV>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<MIN>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<MAXI>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<MED3>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<MIN3>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<CMP>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<CMP64>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<CMP64_CND>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<CND_S>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<ADD_S>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<ADD_LIT>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<ADDCO>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<ADDC>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<MUL_LO>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<MUL24>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<PERM>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<ALIGNBIT>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<BFI>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<LSHL_OR>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<OR3>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<ADD_LSHL>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<FMA>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<MOV_DPP>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<READFL>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<MBCNT>(json, d_out, d_cyc, clock_mhz); json += ",\n";
  run_kind<CVT>(json, d_out, d_cyc, clock_mhz);
  json += "\n ]\n}\n";
  fputs(json.c_str(), stdout);
  return 0;
}
