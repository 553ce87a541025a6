// Micro-benchmark: sustained fp32 VALU rate on gfx950 for plain v_fma_f32 versus
// packed v_pk_fma_f32, at several occupancies.  Decides whether the sweep kernel
// should carry two rays per lane as a float2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

template <int ITERS>
__global__ void __launch_bounds__(256) k_fma(float* out, float a, float b) {
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ITERS>
__global__ void __launch_bounds__(256) k_pkfma(float* out, float a, float b) {
  f2 acc[16];
  f2 va = {a, a * 1.0001f}, vb = {b, b * 0.999f};
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(acc[i], va, vb);
    }
  }
  f2 s = {0, 0};
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// fp64 fma for reference
template <int ITERS>
__global__ void __launch_bounds__(256) k_dfma(double* out, double a, double b) {
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 0.001 + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz lds/blk %zu regs/blk %d\n", p.name, p.multiProcessorCount,
         p.clockRate, p.sharedMemPerBlock, p.regsPerBlock);
  const int ITERS = 4096;
  void* out;
  CHECK(hipMalloc(&out, 256 * 64 * 256 * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int bpc : {1, 2, 4, 8}) {
    int blocks = p.multiProcessorCount * bpc;
    for (int kind = 0; kind < 3; ++kind) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        if (kind == 0) k_fma<ITERS><<<blocks, 256>>>((float*)out, 0.999f, 0.001f);
        else if (kind == 1) k_pkfma<ITERS><<<blocks, 256>>>((float*)out, 0.999f, 0.001f);
        else k_dfma<ITERS><<<blocks, 256>>>((double*)out, 0.999, 0.001);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      double lanes = (double)blocks * 256;
      double flop = lanes * ITERS * 4 * 16 * 2 * (kind == 1 ? 2 : 1);
      printf("blocks/CU %d %-8s %8.3f ms  %8.2f TFLOP/s\n", bpc,
             kind == 0 ? "fma32" : kind == 1 ? "pkfma32" : "fma64", best, flop / best * 1e-9);
    }
  }
  return 0;
}
