// Stand-alone probe (build on the GPU box or here: hipcc --offload-arch=gfx950 -O3 tools/probe_occ1_loop.hip -o probe_occ1):
// what the K loop of the float64 MFMA GEMM can sustain when everything it needs is already in LDS - the inner loop
// only, no global loads, no epilogue - in two shapes:
//   A  the shipped shape: 4 waves x (64 x 64) wave tiles, two workgroups per CU (<= 256 VGPRs), the fragments of a
//      k-step read from LDS right before its 64 MFMAs (no room to double-buffer them)
//   B  one workgroup per CU (512 registers per lane): 4 waves x (128 x 64) wave tiles - 24 fragment reads per 128 MFMAs
//      instead of 20 per 64 - the NEXT k-step's fragments fetched while this one's MFMAs run, the reads placed between
//      the MFMAs with sched_group_barrier
// Answers, before a full kernel is written, whether shape B lifts the ~9 % the LDS fragment reads cost shape A
// ("where a tile's time goes").
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int KC = 16;

// K-strided LDS image [KC][ROWS + 16] (row stride = 16 mod 32 doubles: conflict-free ds_read_b64 for the fragment map)
template <int ROWS> struct Img { static constexpr int LD = ROWS + 16; static constexpr int SIZE = KC * LD; };

template <int MI, int NJ, int WG_PER_CU, bool PIPE, bool BARRIER>
__global__ __launch_bounds__(256, WG_PER_CU) void loop_kernel(double* __restrict__ out, int nchunk) {
  constexpr int BM = 2 * MI * 16, BN = 2 * NJ * 16;  // 2 x 2 waves
  using IA = Img<BM>;
  using IB = Img<BN>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* sA = smem;                   // [2 stages][IA::SIZE]
  double* sB = smem + 2 * IA::SIZE;    // [2 stages][IB::SIZE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int q = tid; q < 2 * IA::SIZE; q += 256) sA[q] = 1.0 + 1e-3 * (q % 97);
  for (int q = tid; q < 2 * IB::SIZE; q += 256) sB[q] = 0.5 - 1e-3 * (q % 89);
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fk = lane >> 4, fblk = (lane >> 2) & 3, fj = lane & 3;
  int brot[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) brot[s] = 4 * ((fblk + s) & 3) + fj;
  double acc[MI][NJ][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
  const double* a0 = sA + fk * IA::LD + wm * (MI * 16) + fr;
  const double* b0 = sB + fk * IB::LD + wn * (NJ * 16);

  if constexpr (!PIPE) {
    for (int c = 0; c < nchunk; ++c) {
      const double* a_s = a0 + (c & 1) * IA::SIZE;
      const double* b_s = b0 + (c & 1) * IB::SIZE;
      for (int s = 0; s < KC / 4; ++s) {
        double af[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = a_s[s * 4 * IA::LD + i * 16];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double bf = b_s[s * 4 * IB::LD + j * 16 + brot[r]];
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[i], bf, acc[i][j][r], 0, 0, 0);
          }
      }
      if (BARRIER) __syncthreads();
    }
  } else {
    double af[2][MI], bf[2][NJ][4];
    auto fetch = [&](int buf, const double* a_s, const double* b_s, int s) {
#pragma unroll
      for (int i = 0; i < MI; ++i) af[buf][i] = a_s[s * 4 * IA::LD + i * 16];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bf[buf][j][r] = b_s[s * 4 * IB::LD + j * 16 + brot[r]];
    };
    fetch(0, a0, b0, 0);
    for (int c = 0; c < nchunk; ++c) {
      const double* a_s = a0 + (c & 1) * IA::SIZE;
      const double* b_s = b0 + (c & 1) * IB::SIZE;
      const double* a_n = a0 + ((c + 1) & 1) * IA::SIZE;
      const double* b_n = b0 + ((c + 1) & 1) * IB::SIZE;
#pragma unroll
      for (int s = 0; s < KC / 4; ++s) {
        const int cur = s & 1;
        if (s + 1 < KC / 4) fetch(cur ^ 1, a_s, b_s, s + 1);
        else fetch(cur ^ 1, a_n, b_n, 0);   // (the next chunk's stage: already resident in this probe)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < MI; ++i)
              acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[cur][i], bf[cur][j][r], acc[i][j][r], 0, 0, 0);
        // one LDS read after every few MFMAs: (MI + 4 NJ) reads among MI * NJ * 4 MFMAs
        constexpr int NREAD = MI + 4 * NJ, NMFMA = MI * NJ * 4, PER = NMFMA / NREAD;
#pragma unroll
        for (int q = 0; q < NREAD; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - PER * NREAD, 0);
      }
      if (BARRIER) __syncthreads();
    }
  }
  double sum = 0.0;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
  out[(size_t)blockIdx.x * 256 + tid] = sum;
}

template <int MI, int NJ, int WG_PER_CU, bool PIPE, bool BARRIER>
int run(const char* name, double* d_out, int nchunk, int ncu) {
  constexpr int BM = 2 * MI * 16, BN = 2 * NJ * 16;
  const size_t lds = (size_t)2 * (Img<BM>::SIZE + Img<BN>::SIZE) * sizeof(double);
  auto kern = loop_kernel<MI, NJ, WG_PER_CU, PIPE, BARRIER>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = ncu * WG_PER_CU * 4;  // four rounds of resident workgroups
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, nullptr, d_out, nchunk / 8);
  CHECK(hipEventRecord(a, nullptr));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, nullptr, d_out, nchunk);
  CHECK(hipEventRecord(b, nullptr));
  CHECK(hipEventSynchronize(b));
  CHECK(hipGetLastError());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const double flops = (double)grid * 2.0 * BM * BN * KC * nchunk;
  printf("%-58s tile %3d x %3d, %d WG/CU, LDS %6zu B: %7.2f TFLOP/s  (%.3f ms)\n", name, BM, BN, WG_PER_CU, lds,
         flops / (ms * 1e-3) / 1e12, ms);
  fflush(stdout);
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("%s, %d CUs\n", prop.name, ncu);
  double* d_out;
  CHECK(hipMalloc(&d_out, sizeof(double) * 256 * ncu * 2 * 4));
  const int nchunk = 2048;
  for (int rep = 0; rep < 2; ++rep) {
    if (run<4, 4, 2, false, true>("A  64x64 wave tile, reads before MFMAs, barrier", d_out, nchunk, ncu)) return 1;
    if (run<4, 4, 2, false, false>("A  64x64 wave tile, reads before MFMAs, no barrier", d_out, nchunk, ncu)) return 1;
    if (run<4, 4, 1, true, true>("   64x64 wave tile, 1 WG/CU, pipelined reads, barrier", d_out, nchunk, ncu)) return 1;
    if (run<8, 4, 1, false, true>("   128x64 wave tile, 1 WG/CU, reads before MFMAs, barrier", d_out, nchunk, ncu)) return 1;
    if (run<8, 4, 1, true, true>("B  128x64 wave tile, 1 WG/CU, pipelined reads, barrier", d_out, nchunk, ncu)) return 1;
    if (run<8, 4, 1, true, false>("B  128x64 wave tile, 1 WG/CU, pipelined reads, no barrier", d_out, nchunk, ncu)) return 1;
  }
  hipFree(d_out);
  return 0;
}
