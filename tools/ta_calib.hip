// ta_calib.hip -- how many per-lane record fetches per nanosecond does a CU's vector-memory path (TA / TCP / L2) sustain for the traversal's
// access pattern?  k_trace_dyn fetches, per lane and step, ONE 80-byte node at a random index as 5 x global_load_dwordx4: 64 lanes -> 64 different
// cache lines per instruction.  If that path is the limiter (and not VALU issue, see tools/valu_calib.hip) the levers are the number of load
// instructions per record and how many distinct lines one instruction touches -- this tool prices them.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ta_calib tools/ta_calib.hip && /tmp/ta_calib      -> one line per (pattern, working set)
//
// Every kernel: 256 CUs x 8 blocks x 256 threads (the traversal kernel's residency), each lane fetches `iters` records at hashed indices and folds them
// into a checksum; time from HIP events.  Patterns:
//   lane<P>x16 @S   per-lane fetch of P x 16 bytes from a record of stride S bytes (80-B nodes: P=5,S=80; 64-B nodes: P=4,S=64; triangles: P=3,S=64; line nodes: P=5,S=128)
//   coop5 @80       the same 64 records per wave, but lane i loads 16-byte piece (i + 64k) % 5 of record (i + 64k) / 5: consecutive lanes read consecutive
//                   addresses; the pieces are exchanged through LDS (5 KiB per wave) and every lane then reads its own record back (ds_read_b128 x 5)
//   dep             each record's index depends on the previous record's data (pointer chase: the latency a single walk sees)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t v)
{
  uint32_t st = v * 747796405u + 2891336453u;
  uint32_t w = ((st >> ((st >> 28) + 4u)) ^ st) * 277803737u;
  return (w >> 22) ^ w;
}

template <int P, bool DEP>
__global__ __launch_bounds__(256) void k_lane(const uint4* __restrict__ src, uint32_t records, uint32_t strideU4, uint32_t iters, uint32_t* sink)
{
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0u, idx = pcg(gid) % records;
  for (uint32_t it = 0; it < iters; it++) {
    const uint4* p = src + (size_t)idx * strideU4;
    uint4 v[P];
#pragma unroll
    for (int k = 0; k < P; k++) v[k] = p[k];
    uint32_t x = 0u;
#pragma unroll
    for (int k = 0; k < P; k++) x ^= v[k].x ^ v[k].w;
    acc ^= x;
    idx = DEP ? (pcg(gid + it) ^ x) % records : pcg(gid * 977u + it * 131071u + 1u) % records;
  }
  if (acc == 0x12345678u) *sink = acc;
}

// cooperative form: 64 records of 80 bytes per wave and iteration through a 5 KiB LDS stage
__global__ __launch_bounds__(256) void k_coop5(const uint4* __restrict__ src, uint32_t records, uint32_t iters, uint32_t* sink)
{
  __shared__ uint4 stage[4][64 * 5];
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t acc = 0u;
  for (uint32_t it = 0; it < iters; it++) {
    const uint32_t idx = pcg(gid * 977u + it * 131071u + 1u) % records; // the record THIS lane wants
    uint4 piece[5];
#pragma unroll
    for (uint32_t k = 0; k < 5u; k++) {
      const uint32_t flat = k * 64u + lane, owner = flat / 5u, part = flat - owner * 5u;
      const uint32_t oidx = (uint32_t)__shfl((int)idx, (int)owner);
      piece[k] = src[(size_t)oidx * 5u + part];
    }
#pragma unroll
    for (uint32_t k = 0; k < 5u; k++) stage[wave][k * 64u + lane] = piece[k];
    __builtin_amdgcn_wave_barrier();
    uint32_t x = 0u;
#pragma unroll
    for (uint32_t k = 0; k < 5u; k++) { const uint4 v = stage[wave][lane * 5u + k]; x ^= v.x ^ v.w; }
    __builtin_amdgcn_wave_barrier();
    acc ^= x;
  }
  if (acc == 0x12345678u) *sink = acc;
}

template <class F>
static void timeit(const char* name, size_t wsBytes, uint32_t recordBytes, uint32_t fetchBytes, uint32_t iters, int cus, F launch)
{
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(8u); CHECK(hipDeviceSynchronize()); // warm
  CHECK(hipEventRecord(e0, 0)); launch(iters); CHECK(hipEventRecord(e1, 0)); CHECK(hipDeviceSynchronize());
  float ms = 0.0f; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double lanes = (double)cus * 8.0 * 256.0, fetches = lanes * iters;
  printf("TA_CALIB pattern=\"%s\" working_set_MB=%.1f record_B=%u fetched_B_per_record=%u fetches=%.0f ms=%.3f fetches_per_ns_per_CU=%.4f wave_fetch_groups_per_us_per_CU=%.3f GBps_useful=%.1f\n",
         name, wsBytes / 1048576.0, recordBytes, fetchBytes, fetches, ms, fetches / (ms * 1e6) / cus, fetches / 64.0 / (ms * 1e3) / cus, fetches * fetchBytes / (ms * 1e6));
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main()
{
  CHECK(hipSetDevice(0));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t big = (size_t)2 << 30;
  uint4* buf; uint32_t* sink;
  CHECK(hipMalloc(&buf, big)); CHECK(hipMalloc(&sink, 4)); CHECK(hipMemset(buf, 1, big)); CHECK(hipMemset(sink, 0, 4));
  const dim3 grid(cus * 8), block(256);
  const size_t sets[] = {(size_t)1 << 20, (size_t)20 << 20, (size_t)376 << 20, big}; // 1 MB (L2), 20 MB (C3's nodes), 376 MB (C4's flat BVH: beyond the Infinity Cache), 2 GiB
  for (size_t ws : sets) {
    const uint32_t it = ws <= ((size_t)20 << 20) ? 256u : 96u;
#define LANE(P, S, NAME) timeit(NAME, ws, S, P * 16u, it, cus, [&](uint32_t iters) { hipLaunchKernelGGL((k_lane<P, false>), grid, block, 0, 0, buf, (uint32_t)(ws / S), S / 16u, iters, sink); })
    LANE(5, 80u, "lane 5x16 @80 (node, today)");
    LANE(4, 64u, "lane 4x16 @64 (a 64-byte node)");
    LANE(5, 128u, "lane 5x16 @128 (node per line)");
    LANE(3, 64u, "lane 3x16 @64 (triangle, 48 of 64 B)");
    LANE(2, 64u, "lane 2x16 @64");
    LANE(1, 64u, "lane 1x16 @64");
    LANE(1, 16u, "lane 1x16 @16");
    timeit("coop 5x16 @80 through LDS", ws, 80u, 80u, it, cus, [&](uint32_t iters) { hipLaunchKernelGGL(k_coop5, grid, block, 0, 0, buf, (uint32_t)(ws / 80u), iters, sink); });
    timeit("lane 5x16 @80, dependent (pointer chase)", ws, 80u, 80u, it / 4u, cus, [&](uint32_t iters) { hipLaunchKernelGGL((k_lane<5, true>), grid, block, 0, 0, buf, (uint32_t)(ws / 80u), 5u, iters, sink); });
  }
  CHECK(hipFree(buf)); CHECK(hipFree(sink));
  return 0;
}
