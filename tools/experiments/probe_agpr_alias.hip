// Stand-alone probe (hipcc --offload-arch=gfx950 -O3 tools/experiments/probe_agpr_alias.hip -o /tmp/probe_agpr):
// with accum_offset = 128 (128 architectural VGPRs + 128 AGPRs), does an instruction that writes v[124:127] disturb
// a[0:3]?  (Round 4: the TOUT = 2 GEMM kernel showed a[0:3] holding what a ds_read2_b64 had put into v[124:127].)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CLOB_A "a0","a1","a2","a3","a4","a5","a6","a7","a120","a121","a122","a123","a124","a125","a126","a127"
#define CLOB_V "v120","v121","v122","v123","v124","v125","v126","v127"
__global__ __launch_bounds__(256, 2) void k(const unsigned* __restrict__ in, unsigned* __restrict__ out, int mode) {
  __shared__ unsigned sm[4096];
  for (int q = threadIdx.x; q < 4096; q += 256) sm[q] = 0x5000000u + q;
  __syncthreads();
  const unsigned lds = (unsigned)(size_t)(sm) + threadIdx.x * 16;
  const unsigned* gp = in + threadIdx.x * 4;
  asm volatile("v_mov_b32 v120, 0xa0\n\tv_mov_b32 v121, 0xa1\n\tv_mov_b32 v122, 0xa2\n\tv_mov_b32 v123, 0xa3\n\tv_mov_b32 v124, 0xa4\n\tv_mov_b32 v125, 0xaf\n\ts_nop 4\n\t"
               "v_accvgpr_write_b32 a0, v120\n\tv_accvgpr_write_b32 a1, v121\n\tv_accvgpr_write_b32 a2, v122\n\tv_accvgpr_write_b32 a3, v123\n\t"
               "v_accvgpr_write_b32 a4, v124\n\tv_accvgpr_write_b32 a127, v125\n\ts_nop 4\n\t"
               "v_mov_b32 v120, 0x120\n\tv_mov_b32 v121, 0x121\n\tv_mov_b32 v122, 0x122\n\tv_mov_b32 v123, 0x123\n\t"
               "v_mov_b32 v124, 0x124\n\tv_mov_b32 v125, 0x125\n\tv_mov_b32 v126, 0x126\n\tv_mov_b32 v127, 0x127\n\ts_nop 4" ::: CLOB_A, CLOB_V);
  if (mode == 1) asm volatile("ds_read2_b64 v[124:127], %0 offset1:130\n\ts_waitcnt lgkmcnt(0)" :: "v"(lds) : "memory", CLOB_V);
  if (mode == 2) asm volatile("ds_read_b128 v[124:127], %0\n\ts_waitcnt lgkmcnt(0)" :: "v"(lds) : "memory", CLOB_V);
  if (mode == 3) asm volatile("global_load_dwordx4 v[124:127], %0, off\n\ts_waitcnt vmcnt(0)" :: "v"(gp) : "memory", CLOB_V);
  if (mode == 4) asm volatile("ds_read2_b64 v[124:127], %0 offset1:130\n\ts_waitcnt lgkmcnt(0)\n\tglobal_store_dwordx4 %1, v[124:127], off\n\ts_nop 2" :: "v"(lds), "v"(out + 4096 + threadIdx.x * 4) : "memory", CLOB_V);
  if (mode == 5) asm volatile("v_mov_b64 v[126:127], v[124:125]" ::: CLOB_V);
  unsigned r[8];
  asm volatile("s_nop 4\n\tv_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3\n\t"
               "v_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a127\n\tv_mov_b32 %6, v124\n\tv_mov_b32 %7, v127"
               : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]));
  for (int q = 0; q < 8; ++q) out[threadIdx.x * 8 + q] = r[q];
}
int main() {
  unsigned *in, *out;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 8192 * 4);
  std::vector<unsigned> h(4096);
  for (int q = 0; q < 4096; ++q) h[q] = 0x7000000u + q;
  hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode <= 5; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, in, out, mode);
    std::vector<unsigned> o(2048);
    if (hipMemcpy(o.data(), out, 2048 * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
    printf("mode %d lane 0: a0..a4 = %x %x %x %x %x  a127 = %x  v124 = %x v127 = %x   (lane 70: a0 = %x, v124 = %x)\n", mode, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[70 * 8], o[70 * 8 + 6]);
  }
  return 0;
}
