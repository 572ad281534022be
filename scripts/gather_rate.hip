// What a per-lane gather costs on gfx950's vector-memory path (round 4: is project_kernel bound by the texture addresser /
// L1 rather than by VALU issue or HBM?).  Build: hipcc --offload-arch=gfx950 -O2 -o scripts/gather_rate scripts/gather_rate.hip
// Run on the GPU box: scripts/gather_rate > gpurun_out/gather_rate.json
//
// Every wave issues ITER x 8 independent loads of WIDTH dwords per lane (eight in flight, so latency does not bind with
// 8 waves per SIMD resident); an instruction's 64 lanes fall into DISTINCT groups, each group reads one random 128-byte
// line of a FOOT-byte footprint (16 KB: L1 hits; 4 MB: L2 hits; 2 GB: HBM), the lanes of a group at consecutive
// WIDTH-dword offsets inside the line (wrapping).  Whole-chip launch (2048 blocks x 256 threads = 8 waves per SIMD), wall
// time by hipEvents -> wave-instructions per clock per CU at the 2.4 GHz shader clock, and bytes returned per clock per CU.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 256;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int WIDTH>
__global__ __launch_bounds__(256) void gather(const uint32_t *__restrict__ buf, uint32_t line_mask, uint32_t distinct, uint32_t *out) {
  const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
  const uint32_t per = 64u / distinct, grp = lane / per, within = lane % per;
  const uint32_t inner = (within * WIDTH) & 31u;  // dword offset inside the 128-byte line
  uint32_t acc = 0;
  uint32_t seed = mix(wave * 64u + grp + 1u);
  for (int it = 0; it < ITER; it++) {
    uint32_t a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      seed = seed * 1664525u + 1013904223u;
      a[k] = ((seed >> 7) & line_mask) * 32u + inner;
    }
    if (WIDTH == 4) {
      uint4 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = *reinterpret_cast<const uint4 *>(buf + a[k]);
#pragma unroll
      for (int k = 0; k < 8; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    } else if (WIDTH == 2) {
      uint2 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = *reinterpret_cast<const uint2 *>(buf + a[k]);
#pragma unroll
      for (int k = 0; k < 8; k++) acc ^= v[k].x ^ v[k].y;
    } else {
      uint32_t v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = buf[a[k]];
#pragma unroll
      for (int k = 0; k < 8; k++) acc ^= v[k];
    }
  }
  if (acc == 0x12345678u) out[0] = acc;  // (keeps the loads)
}

int main() {
  CHECK(hipSetDevice(0));
  const size_t max_bytes = 2ull << 30;
  uint32_t *buf, *out;
  CHECK(hipMalloc(&buf, max_bytes));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(buf, 1, max_bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int blocks = 2048;
  const double clk = 2.4e9, cus = 256;
  const size_t foots[3] = {16u << 10, 4u << 20, (size_t)2u << 30};
  const char *foot_name[3] = {"16KB (L1 hits)", "4MB (L2 hits)", "2GB (HBM)"};
  printf("{\n \"device\": \"gfx950\", \"blocks\": %d, \"iter\": %d, \"loads_in_flight_per_wave\": 8, \"results\": [\n", blocks, ITER);
  bool first = true;
  for (int w = 0; w < 3; w++) {
    const int width = w == 0 ? 1 : w == 1 ? 2 : 4;
    for (int f = 0; f < 3; f++) {
      for (uint32_t distinct = 1; distinct <= 64; distinct *= 4) {
        const uint32_t line_mask = (uint32_t)(foots[f] / 128) - 1u;
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
          CHECK(hipEventRecord(e0));
          if (width == 1) gather<1><<<blocks, 256>>>(buf, line_mask, distinct, out);
          else if (width == 2) gather<2><<<blocks, 256>>>(buf, line_mask, distinct, out);
          else gather<4><<<blocks, 256>>>(buf, line_mask, distinct, out);
          CHECK(hipEventRecord(e1));
          CHECK(hipEventSynchronize(e1));
          float ms;
          CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        const double insts = (double)blocks * 4 * ITER * 8;
        const double cyc = best * 1e-3 * clk;
        printf("%s  {\"width_dwords\": %d, \"footprint\": \"%s\", \"distinct_lines_per_inst\": %u, \"ms\": %.3f, \"clocks_per_inst_per_cu\": %.2f, "
               "\"bytes_per_clock_per_cu\": %.1f, \"lines_per_clock_per_cu\": %.3f}",
               first ? "" : ",\n", width, foot_name[f], distinct, best, cyc * cus / insts, insts * 64 * width * 4 / (cyc * cus), insts * distinct / (cyc * cus));
        first = false;
        fflush(stdout);
      }
    }
  }
  printf("\n ]\n}\n");
  return 0;
}
