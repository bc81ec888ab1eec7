// One-hot probe of v_mfma_f64_4x4x4_4b_f64's cbsz / abid / blgp modifiers: which D lanes does A lane pa x B lane pb reach?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int CB, int AB, int BL>
__global__ void probe(int* table) {
  const int lane = threadIdx.x;
  for (int pa = 0; pa < 64; ++pa)
    for (int pb = 0; pb < 64; ++pb) {
      double a = lane == pa ? 1.0 : 0.0, b = lane == pb ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CB, AB, BL);
      unsigned long long m = __ballot(d != 0.0);
      // encode: lowest hit lane + 64 * (number of hits - 1); sign of the hit in bit 30
      int enc = m ? (__ffsll((long long)m) - 1) + 64 * (__popcll(m) - 1) : -1;
      unsigned long long neg = __ballot(d < 0.0);
      if (m && neg) enc |= 1 << 30;
      if (lane == 0) table[pa * 64 + pb] = enc;
    }
}
template <int CB, int AB, int BL>
void run(int* d, int* h) {
  hipLaunchKernelGGL((probe<CB, AB, BL>), dim3(1), dim3(64), 0, nullptr, d);
  hipDeviceSynchronize();
  hipMemcpy(h, d, sizeof(int) * 4096, hipMemcpyDeviceToHost);
  printf("cbsz=%d abid=%d blgp=%d:\n", CB, AB, BL);
  // A lane (k, blk, i) = 16k + 4blk + i ; B lane (k, blk, j)
  for (int pa : {0, 1, 4, 5, 16}) {
    printf("  A lane %2d pairs:", pa);
    int cnt = 0;
    for (int pb = 0; pb < 64; ++pb)
      if (h[pa * 64 + pb] >= 0) {
        int e = h[pa * 64 + pb];
        printf(" (B%d->D%d%s x%d)", pb, e & 63, (e >> 30) & 1 ? "neg" : "", ((e >> 6) & 0xffffff) % 64 + 1);
        ++cnt;
      }
    printf("  [%d]\n", cnt);
  }
}
int main() {
  int *d, *h = (int*)malloc(sizeof(int) * 4096);
  hipMalloc(&d, sizeof(int) * 4096);
  run<0, 0, 0>(d, h);
  run<1, 0, 0>(d, h);
  run<1, 1, 0>(d, h);
  run<2, 0, 0>(d, h);
  run<2, 1, 0>(d, h);
  run<2, 3, 0>(d, h);
  run<0, 0, 1>(d, h);
  run<0, 0, 2>(d, h);
  run<0, 0, 3>(d, h);
  run<0, 0, 4>(d, h);
  run<0, 0, 5>(d, h);
  run<0, 0, 7>(d, h);
  return 0;
}
