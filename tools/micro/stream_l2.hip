// How fast can ONE workgroup re-read a parameter block that sits in the L2?  (the generic-shape rollout / update stream 170-230 KB of
// weights per step and workgroup: tools/micro/stream_l2.hip, `hipcc -O3 --offload-arch=gfx950 stream_l2.hip -o stream_l2 && ./stream_l2`)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int DEPTH, int AUX, int THREADS>
__global__ void __launch_bounds__(THREADS) stream_kernel(const float* p, int n_floats, int iters, float* out, unsigned long long* cyc) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, n_floats * 4 + 16, 0x00020000);      // (p may be offset by 1..3 floats: the misaligned runs)
  f4 acc = f4{0.f, 0.f, 0.f, 0.f};
  const int tid = threadIdx.x;
  const int per_round = THREADS * 4 * DEPTH;      // floats per round of DEPTH loads per thread
  unsigned long long t0 = 0;
  if (tid == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
    for (int base = 0; base < n_floats; base += per_round) {
      f4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (base + (d * THREADS + tid) * 4) * 4, 0, AUX));
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
    __syncthreads();
  }
  unsigned long long t1 = 0;
  if (tid == 0) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)); cyc[blockIdx.x] = t1 - t0; }
  out[blockIdx.x * THREADS + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int DEPTH, int AUX, int THREADS>
void run(const char* name, float* d_p, int n_floats, int wgs, float* d_out, unsigned long long* d_cyc) {
  const int iters = 200;
  hipLaunchKernelGGL((stream_kernel<DEPTH, AUX, THREADS>), dim3(wgs), dim3(THREADS), 0, 0, d_p, n_floats, iters, d_out, d_cyc);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL((stream_kernel<DEPTH, AUX, THREADS>), dim3(wgs), dim3(THREADS), 0, 0, d_p, n_floats, iters, d_out, d_cyc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> c(wgs);
  hipMemcpy(c.data(), d_cyc, wgs * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto x : c) mean += (double)x; mean /= wgs;
  printf("%-34s %3d WGs x %3d thr, depth %2d: %7.2f us per pass of %d KB (%6.0f clocks of s_memtime) -> %5.1f B/clk per WG, %6.1f GB/s per WG\n", name, wgs, THREADS, DEPTH,
         1e3 * ms / iters, n_floats * 4 / 1024, mean / iters, n_floats * 4.0 / (mean / iters), n_floats * 4.0 / (1e-3 * ms / iters) / 1e9);
}
int main() {
  const int n_floats = 57344;      // 224 KB: three 128-128 branches
  float* d_p; float* d_out; unsigned long long* d_cyc;
  hipMalloc(&d_p, n_floats * 4 + 64); hipMemset(d_p, 0, n_floats * 4 + 64);
  hipMalloc(&d_out, 1024 * 1024 * 4); hipMalloc(&d_cyc, 4096 * 8);
  for (int mis : {1, 2, 3}) {      // 16-byte loads whose addresses are only dword / 8-byte aligned (parameter blocks at arbitrary offsets)
    char nm[64]; snprintf(nm, sizeof nm, "plain loads, base + %d floats", mis);
    run<8, 0, 256>(nm, d_p + mis, n_floats - 4, 1, d_out, d_cyc);
    run<8, 0, 256>(nm, d_p + mis, n_floats - 4, 64, d_out, d_cyc);
  }
  for (int wgs : {1, 8, 64}) {
    run<4, 0, 256>("plain loads", d_p, n_floats, wgs, d_out, d_cyc);
    run<8, 0, 256>("plain loads", d_p, n_floats, wgs, d_out, d_cyc);
    run<16, 0, 256>("plain loads", d_p, n_floats, wgs, d_out, d_cyc);
    run<8, 0, 512>("plain loads", d_p, n_floats, wgs, d_out, d_cyc);
    run<8, 1, 512>("sc0 loads", d_p, n_floats, wgs, d_out, d_cyc);
    run<8, 16, 512>("sc1 loads", d_p, n_floats, wgs, d_out, d_cyc);
    run<16, 16, 512>("sc1 loads", d_p, n_floats, wgs, d_out, d_cyc);
  }
  return 0;
}
