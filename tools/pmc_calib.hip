// pmc_calib.hip -- known-byte-count kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS code base's access
// patterns (MI355X_MICROARCH.md, "HBM": only wide coalesced reads are calibrated there -- reported = 1/2 of the bytes).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/pmc_calib tools/pmc_calib.hip
//   rocprofv3 --pmc FETCH_SIZE -d out/f -o calib -- tools/build/pmc_calib      (and a second pass with --pmc WRITE_SIZE)
//   python tools/pmc_calib_report.py out/f/calib_results.db out/w/calib_results.db  -> profiles/pmc_calibration.json
//
// Every kernel prints the bytes its lanes ask for ("algorithmic") on stdout; the report divides the counter by it.
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

// 1. wide coalesced streaming read: 16 B per lane, consecutive lanes consecutive addresses
__global__ void calib_stream_read16(const uint4* __restrict__ src, size_t n, uint32_t* sink)
{
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;
}
// 2. the path pool's Slot gather: 64 B (4 x dwordx4) per lane at a random 64-B slot of a `slots`-entry pool
__global__ void calib_gather64(const uint4* __restrict__ src, uint32_t slots, uint32_t count, uint32_t* sink)
{
  uint32_t acc = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const uint4* p = src + (size_t)(pcg(i) % slots) * 4u;
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc ^= a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
// 3. the traversal's node fetch: 80 B (5 x dwordx4) per lane at a random node of a `nodes`-entry array
__global__ void calib_node80(const uint4* __restrict__ src, uint32_t nodes, uint32_t count, uint32_t* sink)
{
  uint32_t acc = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const uint4* p = src + (size_t)(pcg(i) % nodes) * 5u;
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    acc ^= a.x ^ b.y ^ c.z ^ d.w ^ e.x;
  }
  if (acc == 0x12345678u) *sink = acc;
}
// 4. wide coalesced streaming write
__global__ void calib_stream_write16(uint4* __restrict__ dst, size_t n)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
// 5. the per-sample colour record: one 16-B store per lane at a random record (partial-line writes)
__global__ void calib_scatter16(uint4* __restrict__ dst, uint32_t records, uint32_t count)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) dst[pcg(i) % records] = make_uint4(i, 1u, 2u, 3u);
}
// 6. k_path's sample records: 16-B stores, runs of 64 consecutive records (one per lane) at a random 1 KiB-aligned place
__global__ void calib_write16_runs(uint4* __restrict__ dst, uint32_t records, uint32_t count)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) dst[(size_t)(pcg(i >> 6) % (records >> 6)) * 64u + (i & 63u)] = make_uint4(i, 1u, 2u, 3u);
}

int main()
{
  CHECK(hipSetDevice(0));
  const size_t big = (size_t)2 << 30; // 2 GiB: beyond the 256 MiB Infinity Cache
  uint4* buf; uint32_t* sink;
  CHECK(hipMalloc(&buf, big)); CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(buf, 1, big)); CHECK(hipMemset(sink, 0, 4));
  const dim3 grid(256 * 8), block(256);
  const uint32_t count = 64u << 20; // 64 Mi lanes' worth of accesses per gather kernel
  for (int rep = 0; rep < 2; rep++) { // (first repetition warms code objects; the report averages both)
    hipLaunchKernelGGL(calib_stream_read16, grid, block, 0, 0, buf, big / 16, sink);
    hipLaunchKernelGGL(calib_gather64, grid, block, 0, 0, buf, (uint32_t)(big / 64), count, sink);              // 2 GiB pool: every gather misses every cache
    hipLaunchKernelGGL(calib_gather64, grid, block, 0, 0, buf, (uint32_t)((268u << 20) / 64), count, sink);     // 268 MB pool (4 Mi slots): Infinity-Cache sized
    hipLaunchKernelGGL(calib_node80, grid, block, 0, 0, buf, (uint32_t)(big / 80), count, sink);                // 2 GiB of nodes
    hipLaunchKernelGGL(calib_node80, grid, block, 0, 0, buf, (uint32_t)((20u << 20) / 80), count, sink);        // 20 MB of nodes (C3's BVH8): L2 / Infinity-Cache resident
    hipLaunchKernelGGL(calib_stream_write16, grid, block, 0, 0, buf, big / 16);
    hipLaunchKernelGGL(calib_scatter16, grid, block, 0, 0, buf, (uint32_t)(big / 16), count);
    hipLaunchKernelGGL(calib_write16_runs, grid, block, 0, 0, buf, (uint32_t)(big / 16), count);
    CHECK(hipDeviceSynchronize());
  }
  // algorithmic bytes per launch, in launch order within a repetition
  printf("calib_stream_read16 %zu\n", big);
  printf("calib_gather64 %zu\ncalib_gather64 %zu\n", (size_t)count * 64, (size_t)count * 64);
  printf("calib_node80 %zu\ncalib_node80 %zu\n", (size_t)count * 80, (size_t)count * 80);
  printf("calib_stream_write16 %zu\ncalib_scatter16 %zu\ncalib_write16_runs %zu\n", big, (size_t)count * 16, (size_t)count * 16);
  CHECK(hipFree(buf)); CHECK(hipFree(sink));
  return 0;
}
