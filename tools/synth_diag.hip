// diag: which stage of the synthetic-row derivation differs between GPU and host IEEE arithmetic?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
__host__ __device__ inline uint64_t mix64(uint64_t z){ z=(z^(z>>30))*0xbf58476d1ce4e5b9ull; z=(z^(z>>27))*0x94d049bb133111ebull; return z^(z>>31);} 
__host__ __device__ inline int synth_v(uint64_t seed, uint64_t idx){ uint64_t h=mix64(seed^(idx*0x9e3779b97f4a7c15ull)); return (int)((h&0xffff)+((h>>16)&0xffff)+((h>>32)&0xffff)+(h>>48))-131070; }
struct Out { long long ss; double sq, scale, prod; float f; unsigned short h, h_direct; };
__global__ void k(Out* o, int d, uint64_t seed, int nrows) {
  int r = blockIdx.x * blockDim.x + threadIdx.x; if (r >= nrows) return;
  long long ss = 0; for (int c = 0; c < d; ++c) { long long v = synth_v(seed, (uint64_t)r*d+c); ss += v*v; }
  double sq = sqrt((double)ss); double scale = 1.0 / sq;
  int v0 = synth_v(seed, (uint64_t)r*d + 5);
  double prod = (double)v0 * scale; float f = (float)prod; 
  _Float16 h = (_Float16)f; _Float16 hd = (_Float16)(float)((double)v0 * (1.0 / sqrt((double)ss)));
  o[r].ss = ss; o[r].sq = sq; o[r].scale = scale; o[r].prod = prod; o[r].f = f; memcpy(&o[r].h, &h, 2); memcpy(&o[r].h_direct, &hd, 2);
}
int main() {
  const int d = 768, n = 100000; uint64_t seed = 3;
  Out* dev; hipMalloc(&dev, n * sizeof(Out)); hipLaunchKernelGGL(k, dim3((n+255)/256), dim3(256), 0, 0, dev, d, seed, n);
  std::vector<Out> g(n); hipMemcpy(g.data(), dev, n*sizeof(Out), hipMemcpyDeviceToHost);
  long bad_ss=0,bad_sq=0,bad_scale=0,bad_prod=0,bad_f=0,bad_h=0,bad_hd=0;
  for (int r = 0; r < n; ++r) {
    long long ss=0; for (int c=0;c<d;++c){ long long v=synth_v(seed,(uint64_t)r*d+c); ss+=v*v; }
    volatile double sq = std::sqrt((double)ss); volatile double scale = 1.0/sq; int v0=synth_v(seed,(uint64_t)r*d+5);
    volatile double prod=(double)v0*scale; volatile float f=(float)prod; _Float16 h=(_Float16)f; unsigned short hb; memcpy(&hb,&h,2);
    bad_ss += ss!=g[r].ss; bad_sq += sq!=g[r].sq; bad_scale += scale!=g[r].scale; bad_prod += prod!=g[r].prod; bad_f += f!=g[r].f; bad_h += hb!=g[r].h; bad_hd += hb!=g[r].h_direct;
    if ((sq!=g[r].sq || scale!=g[r].scale) && bad_sq+bad_scale < 5) printf("row %d ss=%lld sq host %a gpu %a scale host %a gpu %a\n", r, ss, (double)sq, g[r].sq, (double)scale, g[r].scale);
  }
  printf("mismatches of %d rows: ss %ld sqrt %ld scale %ld prod %ld f32 %ld f16 %ld f16(one-expression) %ld\n", n, bad_ss,bad_sq,bad_scale,bad_prod,bad_f,bad_h,bad_hd);
  return 0;
}
