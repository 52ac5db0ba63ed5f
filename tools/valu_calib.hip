// valu_calib.hip -- VALU issue-rate microbenchmark for gfx950: how many cycles does a SIMD need per wave64 VALU instruction?
// (bench.py priced one at 4 cycles, MI355X_MICROARCH.md says 2 -- this settles which, per instruction class.)
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/valu_calib tools/valu_calib.hip
//   tools/build/valu_calib                      -> one line per (instruction, waves/SIMD): cycles per instruction and SIMD
//   rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES -d out -o valu -- tools/build/valu_calib --pmc
//   python tools/valu_calib_report.py <stdout of the plain run> [<results.db of the pmc run>]  -> profiles/valu_issue_calibration.{txt,json}
//
// Method: every wave executes REPS x 64 instructions of one kind over 8 independent accumulators (no dependent-issue stalls: the chain
// distance is 8 instructions) and reads the shader clock (s_memtime) around the loop.  A launch puts exactly W waves on every SIMD
// (grid = CUs x W blocks of 256 threads = 4 waves, one per SIMD; checked from HW_ID), so
//     cycles per instruction and SIMD = wave cycles / (W x instructions per wave)
// is independent of the clock the chip happens to run at; the wall-clock rate (HIP events) is reported next to it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

enum Op : int { OP_FMA = 0, OP_PK_FMA, OP_CVT_UBYTE, OP_MOV, OP_ADD_U32, OP_FMA_DEP, OP_MIX_FMA_CVT, OP_MAX3, OP_MIX_FMA_MAX3, OP_MIX_FMA_MOV, OP_MIX_FMA_ADDU, OP_MIX_CVT_ADDU, OP_MIX_PKFMA_CVT2, OP_COUNT };
static const char* kOpName[OP_COUNT] = {"v_fma_f32", "v_pk_fma_f32", "v_cvt_f32_ubyte0", "v_mov_b32", "v_add_u32", "v_fma_f32 (dependent chain)",
                                        "v_cvt_f32_ubyte0 + v_fma_f32 (1:1)", "v_max3_f32", "v_fma_f32 + v_max3_f32 (1:1)", "v_fma_f32 + v_mov_b32 (1:1)",
                                        "v_fma_f32 + v_add_u32 (1:1)", "v_cvt_f32_ubyte0 + v_add_u32 (1:1)", "v_pk_fma_f32 + 2 v_cvt_f32_ubyte0 (node test)"};

#define X8(S) S S S S S S S S

template <int OP>
__global__ __launch_bounds__(256) void k_valu(uint32_t reps, float seed, unsigned long long* __restrict__ cycles, uint32_t* __restrict__ hwid, float* __restrict__ sink)
{
  float a0 = seed, a1 = seed + 1.0f, a2 = seed + 2.0f, a3 = seed + 3.0f, a4 = seed + 4.0f, a5 = seed + 5.0f, a6 = seed + 6.0f, a7 = seed + 7.0f;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const f2 pm = {0.999f, 1.001f}, pa = {0.5f, -0.5f};
  const float m = 0.999f, c = 0.5f;
  uint32_t u0 = (uint32_t)threadIdx.x, u1 = u0 + 1u, u2 = u0 + 2u, u3 = u0 + 3u, u4 = u0 + 4u, u5 = u0 + 5u, u6 = u0 + 6u, u7 = u0 + 7u;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (uint32_t r = 0; r < reps; r++) {
    if (OP == OP_FMA) {
      asm volatile(X8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                      "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    } else if (OP == OP_PK_FMA) {
      asm volatile(X8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                      "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pa));
    } else if (OP == OP_CVT_UBYTE) {
      asm volatile(X8("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte0 %1, %9\n v_cvt_f32_ubyte0 %2, %10\n v_cvt_f32_ubyte0 %3, %11\n"
                      "v_cvt_f32_ubyte0 %4, %12\n v_cvt_f32_ubyte0 %5, %13\n v_cvt_f32_ubyte0 %6, %14\n v_cvt_f32_ubyte0 %7, %15\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(u4), "v"(u5), "v"(u6), "v"(u7));
    } else if (OP == OP_MOV) {
      asm volatile(X8("v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n v_mov_b32 %2, %10\n v_mov_b32 %3, %11\n"
                      "v_mov_b32 %4, %12\n v_mov_b32 %5, %13\n v_mov_b32 %6, %14\n v_mov_b32 %7, %15\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(u4), "v"(u5), "v"(u6), "v"(u7));
    } else if (OP == OP_ADD_U32) {
      asm volatile(X8("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                      "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(r));
    } else if (OP == OP_FMA_DEP) { // every instruction waits for the one before it: issue + result latency of one wave
      asm volatile(X8("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                      "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n")
                   : "+v"(a0) : "v"(m), "v"(c));
    } else if (OP == OP_MIX_FMA_CVT) { // the node test's mix: a byte conversion feeding an fma
      asm volatile(X8("v_cvt_f32_ubyte0 %0, %8\n v_fma_f32 %1, %1, %12, %13\n v_cvt_f32_ubyte0 %2, %9\n v_fma_f32 %3, %3, %12, %13\n"
                      "v_cvt_f32_ubyte0 %4, %10\n v_fma_f32 %5, %5, %12, %13\n v_cvt_f32_ubyte0 %6, %11\n v_fma_f32 %7, %7, %12, %13\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(m), "v"(c));
    } else if (OP == OP_MIX_FMA_MAX3) {
      asm volatile(X8("v_fma_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                      "v_fma_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    } else if (OP == OP_MIX_FMA_MOV) {
      asm volatile(X8("v_fma_f32 %0, %0, %8, %9\n v_mov_b32 %1, %10\n v_fma_f32 %2, %2, %8, %9\n v_mov_b32 %3, %11\n"
                      "v_fma_f32 %4, %4, %8, %9\n v_mov_b32 %5, %10\n v_fma_f32 %6, %6, %8, %9\n v_mov_b32 %7, %11\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "v"(u0), "v"(u1));
    } else if (OP == OP_MIX_FMA_ADDU) {
      asm volatile(X8("v_fma_f32 %0, %0, %8, %9\n v_add_u32 %4, %4, %10\n v_fma_f32 %1, %1, %8, %9\n v_add_u32 %5, %5, %10\n"
                      "v_fma_f32 %2, %2, %8, %9\n v_add_u32 %6, %6, %10\n v_fma_f32 %3, %3, %8, %9\n v_add_u32 %7, %7, %10\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(m), "v"(c), "v"(r));
    } else if (OP == OP_MIX_CVT_ADDU) {
      asm volatile(X8("v_cvt_f32_ubyte0 %0, %8\n v_add_u32 %4, %4, %10\n v_cvt_f32_ubyte0 %1, %9\n v_add_u32 %5, %5, %10\n"
                      "v_cvt_f32_ubyte0 %2, %8\n v_add_u32 %6, %6, %10\n v_cvt_f32_ubyte0 %3, %9\n v_add_u32 %7, %7, %10\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(u0), "v"(u1), "v"(r));
    } else if (OP == OP_MIX_PKFMA_CVT2) { // the node test's inner pattern: two byte conversions feed one packed fma (12 instructions per group of 4 planes)
      asm volatile(X8("v_cvt_f32_ubyte0 %4, %8\n v_cvt_f32_ubyte1 %5, %8\n v_pk_fma_f32 %0, %0, %10, %11\n v_cvt_f32_ubyte2 %6, %8\n v_cvt_f32_ubyte3 %7, %8\n v_pk_fma_f32 %1, %1, %10, %11\n"
                      "v_cvt_f32_ubyte0 %4, %9\n v_cvt_f32_ubyte1 %5, %9\n v_pk_fma_f32 %2, %2, %10, %11\n v_cvt_f32_ubyte2 %6, %9\n v_cvt_f32_ubyte3 %7, %9\n v_pk_fma_f32 %3, %3, %10, %11\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u0), "v"(u1), "v"(pm), "v"(pa));
    } else if (OP == OP_MAX3) {
      asm volatile(X8("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                      "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63u) == 0u) {
    cycles[wave] = t1 - t0;
    uint32_t id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    hwid[wave] = id;
  }
  const float s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)) + (p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y) + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
  if (s == 0.12345f) *sink = s; // keeps the accumulators alive
}

template <int OP>
static void run(int cus, int W, uint32_t reps, unsigned long long* dCycles, uint32_t* dHw, float* dSink, bool quiet)
{
  const int blocks = cus * W, waves = blocks * 4;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_valu<OP>, dim3(blocks), dim3(256), 0, 0, 64u, 1.0f, dCycles, dHw, dSink); // warm (code object, clocks)
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_valu<OP>, dim3(blocks), dim3(256), 0, 0, reps, 1.0f, dCycles, dHw, dSink);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0.0f; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> cyc(waves); std::vector<uint32_t> hw(waves);
  CHECK(hipMemcpy(cyc.data(), dCycles, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hw.data(), dHw, waves * sizeof(uint32_t), hipMemcpyDeviceToHost));
  std::sort(cyc.begin(), cyc.end());
  const double instrPerWave = (double)reps * (OP == OP_MIX_PKFMA_CVT2 ? 96.0 : 64.0), med = (double)cyc[waves / 2], mx = (double)cyc[waves - 1];
  // placement check: waves per (se, cu, simd) -- HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se_id[15:13] ... (gfx9 layout; xcc is not in HW_ID,
  // so distinct (se, sh, cu, simd) tuples are counted per XCD-agnostic key and the maximum share is reported instead of asserted)
  std::vector<uint32_t> keys(waves);
  for (int i = 0; i < waves; i++) keys[i] = hw[i] & 0xfff0u & ~0xc0u;
  std::sort(keys.begin(), keys.end());
  int maxShare = 0, runLen = 0;
  for (int i = 0; i < waves; i++) { runLen = (i && keys[i] == keys[i - 1]) ? runLen + 1 : 1; maxShare = std::max(maxShare, runLen); }
  const double simds = (double)cus * 4.0;
  const double cycPerInstrSimd = med / ((double)W * instrPerWave);
  const double wallRate = (double)waves * instrPerWave / (ms * 1e-3) / simds; // wave-instructions per second and SIMD
  if (!quiet)
    printf("VALU_CALIB op=\"%s\" waves_per_simd=%d instr_per_wave=%.0f median_wave_cycles=%.0f max_wave_cycles=%.0f cycles_per_instr_per_simd=%.4f wall_ms=%.4f "
           "wall_instr_per_simd_per_ns=%.5f implied_clock_GHz_at_that_rate=%.4f max_waves_sharing_hw_key=%d\n",
           kOpName[OP], W, instrPerWave, med, mx, cycPerInstrSimd, ms, wallRate * 1e-9, wallRate * cycPerInstrSimd * 1e-9, maxShare);
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv)
{
  const bool pmc = argc > 1 && !strcmp(argv[1], "--pmc"); // counter collection serialises and slows the kernels: fewer repetitions, only the 8-wave rows matter
  CHECK(hipSetDevice(0));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("VALU_CALIB_DEVICE name=\"%s\" cus=%d clock_khz=%d\n", prop.name, cus, prop.clockRate);
  unsigned long long* dCycles; uint32_t* dHw; float* dSink;
  const int maxWaves = cus * 8 * 4;
  CHECK(hipMalloc(&dCycles, maxWaves * sizeof(unsigned long long))); CHECK(hipMalloc(&dHw, maxWaves * sizeof(uint32_t))); CHECK(hipMalloc(&dSink, 4));
  const uint32_t reps = pmc ? 2000u : 20000u; // x 64 instructions per wave
  const int Ws[] = {1, 2, 4, 8};
  for (int wi = 0; wi < 4; wi++) {
    const int W = Ws[wi];
    if (pmc && W != 8 && W != 1) continue;
    run<OP_FMA>(cus, W, reps, dCycles, dHw, dSink, false);
    run<OP_PK_FMA>(cus, W, reps, dCycles, dHw, dSink, false);
    run<OP_CVT_UBYTE>(cus, W, reps, dCycles, dHw, dSink, false);
    run<OP_MOV>(cus, W, reps, dCycles, dHw, dSink, false);
    run<OP_ADD_U32>(cus, W, reps, dCycles, dHw, dSink, false);
    run<OP_MIX_FMA_CVT>(cus, W, reps, dCycles, dHw, dSink, false);
    run<OP_MAX3>(cus, W, reps, dCycles, dHw, dSink, false);
    run<OP_MIX_FMA_MAX3>(cus, W, reps, dCycles, dHw, dSink, false);
    run<OP_MIX_FMA_MOV>(cus, W, reps, dCycles, dHw, dSink, false);
    run<OP_MIX_FMA_ADDU>(cus, W, reps, dCycles, dHw, dSink, false);
    run<OP_MIX_CVT_ADDU>(cus, W, reps, dCycles, dHw, dSink, false);
    run<OP_MIX_PKFMA_CVT2>(cus, W, reps, dCycles, dHw, dSink, false);
    if (W == 1) run<OP_FMA_DEP>(cus, W, reps / 4u, dCycles, dHw, dSink, false);
  }
  CHECK(hipFree(dCycles)); CHECK(hipFree(dHw)); CHECK(hipFree(dSink));
  return 0;
}
