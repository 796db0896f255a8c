// mfma_sum_test.hip -- the fp64 matrix core as a cross-lane adder: sum of 64 lanes' doubles in two v_mfma_f64_16x16x4 with
// B = 1 (row sums over k, then over the four lane groups) + 3 VALU adds, instead of an 18-instruction DPP butterfly.
// Checks the lane layout assumption on the device against a host sum.  Measured in the sweep (every wave sum replaced): 1028 ->
// 960 us for the spilling 10-pair shape at 10000 x 5 x 10000, but +2 % .. +9 % at every smaller size (the two 8-pass MFMAs
// and their wait states sit between pass 1 and pass 2) and no gain for the non-spilling shapes -- not used by the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double v4f64_t __attribute__((ext_vector_type(4)));
// inline assembly so that the results land in VGPRs (the compiler's choice for the builtin is AGPRs + 8 v_accvgpr_read);
// the s_nop's are the wait states the ISA requires around an 8-pass fp64 MFMA
__device__ __forceinline__ v4f64_t mfma_rowsums(double a) {
  v4f64_t d;
  const double one = 1.0;   // (source B must be a register)
  asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, 0\n\ts_nop 15\n\ts_nop 2" : "=&v"(d) : "v"(a), "v"(one));
  return d;
}
__device__ __forceinline__ double wave_sum_mfma(double v) {
  const v4f64_t d = mfma_rowsums(v);
  const double p = (d[0] + d[1]) + (d[2] + d[3]);
  const v4f64_t e = mfma_rowsums(p);
  return e[0];
}
__global__ void k(const double *in, double *out) {
  const double s = wave_sum_mfma(in[blockIdx.x * 64 + threadIdx.x]);
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
  const int nb = 64;
  std::vector<double> h(nb * 64), r(nb * 64);
  unsigned long long st = 88172645463325252ull;
  for (auto &v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (double)(st >> 11) / 9007199254740992.0 * (1 + (st & 7)) - 0.3; }
  for (int i = 0; i < 64; i++) h[i] = (double)(1ull << (i % 40)) * (i + 1);   // block 0: exactly representable sums
  double *din, *dout;
  hipMalloc(&din, h.size() * 8); hipMalloc(&dout, h.size() * 8);
  hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, din, dout);
  hipMemcpy(r.data(), dout, r.size() * 8, hipMemcpyDeviceToHost);
  int bad = 0; double worst = 0;
  for (int b = 0; b < nb; b++) {
    long double ref = 0; for (int i = 0; i < 64; i++) ref += h[b * 64 + i];
    for (int i = 0; i < 64; i++) {
      const double rel = std::fabs((double)((long double)r[b * 64 + i] - ref) / (double)ref);
      if (r[b * 64 + i] != r[b * 64]) bad++;          // every lane holds the same total
      if (rel > worst) worst = rel;
    }
  }
  printf("lanes disagreeing: %d, worst relative error vs long double sum: %.3g (block 0 exact: %s)\n", bad, worst,
         r[0] == [&] { double s = 0; for (int i = 0; i < 64; i++) s += h[i]; return s; }() ? "yes" : "no");
  return bad != 0 || worst > 1e-14;
}
