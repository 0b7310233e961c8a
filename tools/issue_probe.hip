// issue_probe.hip -- round 4 probes for the accumulation pass (gfx950):
//   A. wave64 issue rate of the integer VALU instructions the staging code can be written in (which are 2-cycle, which 4-cycle);
//   B. do an MFMA-only wave and a VALU-only wave on the SAME SIMD overlap (time = max) or serialise (time = sum)?
//      and how many VALU instructions fit between the MFMAs of ONE wave for free?
//   C. LDS stores: ds_write_b64 / b128 aligned against byte-misaligned.
// Build: hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o tools/issue_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ------------------------------------------------------------------ A: issue rates
#define NACC 32
#define ITERS 1024
#define OPA(n, txt) if (OP == n) asm volatile(txt : "+v"(acc[i]) : "v"(a), "v"(b))
template <int OP>
__global__ __launch_bounds__(256) void rate(int *out, int a0, int b0) {
  int acc[NACC];
  int a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      OPA(0, "v_and_b32 %0, %1, %0");
      OPA(1, "v_or_b32 %0, %1, %0");
      OPA(2, "v_xor_b32 %0, %1, %0");
      OPA(3, "v_add_u32 %0, %1, %0");
      OPA(4, "v_sub_u32 %0, %1, %0");
      OPA(5, "v_lshlrev_b32 %0, 3, %0");
      OPA(6, "v_lshrrev_b32 %0, 3, %0");
      OPA(7, "v_ashrrev_i32 %0, 3, %0");
      OPA(8, "v_mov_b32 %0, %1");
      OPA(9, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf");
      OPA(10, "v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf");
      OPA(11, "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf");
      OPA(12, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
      OPA(13, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf");
      OPA(14, "v_cndmask_b32 %0, %1, %0, vcc");
      OPA(15, "v_and_or_b32 %0, %0, %1, %2");
      OPA(16, "v_or3_b32 %0, %0, %1, %2");
      OPA(17, "v_lshl_or_b32 %0, %0, 3, %1");
      OPA(18, "v_xnor_b32 %0, %1, %0");
      OPA(19, "v_not_b32 %0, %0");
      OPA(20, "v_add_u16 %0, %1, %0");
      OPA(21, "v_sub_u16 %0, %1, %0");
      OPA(22, "v_mul_u32_u24 %0, %1, %0");
      OPA(23, "v_mad_u32_u24 %0, %1, %2, %0");
      OPA(24, "v_perm_b32 %0, %1, %0, %2");
      OPA(25, "v_alignbyte_b32 %0, %1, %0, 1");
      OPA(26, "v_pk_sub_i16 %0, %1, %0");
      OPA(27, "v_dot4_i32_i8 %0, %1, %2, %0");
      OPA(28, "v_bfi_b32 %0, %1, %2, %0");
      OPA(29, "v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD");
      OPA(30, "v_max_u32 %0, %1, %0");
      OPA(31, "v_min_u32 %0, %1, %0");
      OPA(32, "v_xad_u32 %0, %0, %1, %2");
      OPA(33, "v_add_lshl_u32 %0, %0, %1, 2");
      OPA(34, "v_lshrrev_b16 %0, 3, %0");
      OPA(35, "v_pk_add_u16 %0, %1, %0");
      OPA(36, "v_sad_u8 %0, %1, %2, %0");
      OPA(37, "v_subrev_u32 %0, %1, %0");
      OPA(38, "v_lshrrev_b64 %0, 3, %0");  // placeholder, replaced below
      OPA(39, "v_mul_lo_u32 %0, %1, %0");
      OPA(40, "v_dot4c_i32_i8 %0, %1, %2");
      OPA(41, "v_dot4_u32_u8 %0, %1, %2, %0");
      OPA(42, "v_dot2c_i32_i16 %0, %1, %2");
      OPA(43, "v_dot8c_i32_i4 %0, %1, %2");
      OPA(44, "v_mac_f32 %0, %1, %2");
      OPA(45, "v_fmac_f32 %0, %1, %2");
    }
    a += it;
  }
  int s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(64) void f64_chain(double *out, double x) {
  double s = (double)threadIdx.x;
  for (int i = 0; i < (1 << 17) / 16; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(s) : "v"(x));
  }
  out[threadIdx.x] = s;
}
template <int OP>
void run_rate(const char *name) {
  int *d;
  const int blocks = 256 * 8;
  CK(hipMalloc(&d, blocks * 256 * sizeof(int)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  rate<OP><<<blocks, 256>>>(d, 3, 5);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  rate<OP><<<blocks, 256>>>(d, 3, 5);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double instr = (double)blocks * 256 * ITERS * NACC;
  const double gips = instr / (ms * 1e-3) / 1e12;
  printf("A  %-28s %8.3f ms  %7.2f T lane-instr/s  (%.2f of 78.6: ~%.1f cycles per wave64 instruction)\n", name, ms, gips, gips / 78.6, 2.0 * 78.6 / gips);
  CK(hipFree(d));
}

// ------------------------------------------------------------------ B: MFMA beside VALU
// MODE 0: every wave MFMA-only; 1: every wave VALU-only; 2: waves (w >> 2) & 1 ? VALU : MFMA (the same SIMD holds both kinds:
// a workgroup's waves go to the SIMDs round-robin); 3: every wave interleaves NV VALU after each MFMA.
// Work per wave: NM MFMAs and / or NVAL VALU instructions per loop trip.
template <int MODE, int NV>
__global__ __launch_bounds__(1024) void coissue(const int *in, int *out, int rounds, int waves_mfma_only) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v4i a = {in[lane], in[lane + 64], in[lane + 128], in[lane + 192]};
  v4i acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
  int v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = in[lane + i];
  const bool valu_role = MODE == 1 || (MODE == 2 && ((wave >> 2) & 1));
  const bool mfma_role = MODE == 0 || (MODE == 2 && !((wave >> 2) & 1));
  if (MODE == 3) {
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, acc0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_alignbyte_b32 %0, %1, %0, 1" : "+v"(v[i & 7]) : "v"(a.x));
        acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, acc1, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_alignbyte_b32 %0, %1, %0, 1" : "+v"(v[i & 7]) : "v"(a.y));
        acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, acc2, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_alignbyte_b32 %0, %1, %0, 1" : "+v"(v[i & 7]) : "v"(a.z));
        acc3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, acc3, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_alignbyte_b32 %0, %1, %0, 1" : "+v"(v[i & 7]) : "v"(a.w));
      }
    }
  } else if (mfma_role) {
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, acc3, 0, 0, 0);
      }
    }
  } else if (valu_role) {
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
      for (int r = 0; r < 16 * NV; ++r) asm volatile("v_alignbyte_b32 %0, %1, %0, 1" : "+v"(v[r & 7]) : "v"(a.x));
    }
  }
  int s = acc0.x ^ acc1.y ^ acc2.z ^ acc3.w;
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int NV>
float run_co(int waves, int *d_in, int *d_out, const char *what) {
  const int rounds = 2000, blocks = 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((coissue<MODE, NV>), dim3(blocks), dim3(64 * waves), 0, 0, d_in, d_out, rounds, 0);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((coissue<MODE, NV>), dim3(blocks), dim3(64 * waves), 0, 0, d_in, d_out, rounds, 0);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // per wave and round: 16 MFMAs and / or 16 * NV VALU
  printf("B  %-64s waves/CU %2d  NV %2d: %8.3f ms  = %6.1f ns per round\n", what, waves, NV, ms, ms * 1e6 / rounds);
  return ms;
}

// ------------------------------------------------------------------ C: LDS stores
template <int BYTES, int MIS>
__global__ __launch_bounds__(256) void ldsw(int *out, int rounds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  unsigned char *p = smem + tid * BYTES * 2 + MIS;  // (rows of 2 * BYTES per lane: distinct banks as far as possible)
  int x = tid;
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      unsigned char *q = p + r * 256 * BYTES * 2;
      if (BYTES == 8) asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(uintptr_t)q), "v"((unsigned long long)x) : "memory");
      if (BYTES == 16) {
        v4i val = {x, x, x, x};
        asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(uintptr_t)q), "v"(val) : "memory");
      }
      if (BYTES == 4) asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(uintptr_t)q), "v"(x) : "memory");
    }
    x += it;
  }
  __syncthreads();
  out[blockIdx.x * 256 + tid] = smem[tid * 4];
}
template <int BYTES, int MIS>
void run_lds(const char *name) {
  int *d;
  const int blocks = 256 * 4, rounds = 2000;
  CK(hipMalloc(&d, blocks * 256 * sizeof(int)));
  const size_t lds = 8 * 256 * BYTES * 2 + 64;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((ldsw<BYTES, MIS>), dim3(blocks), dim3(256), lds, 0, d, rounds);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((ldsw<BYTES, MIS>), dim3(blocks), dim3(256), lds, 0, d, rounds);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // 4 workgroups a CU in one round (1024 blocks over 256 CUs): per CU 4 x 4 waves x rounds x 8 stores
  const double stores_per_cu = 4.0 * 4 * rounds * 8;
  printf("C  %-40s %8.3f ms  %6.1f ns per wave-store per CU (= %5.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / stores_per_cu, ms * 1e6 / stores_per_cu * 2.4);
  CK(hipFree(d));
}

int main(int argc, char **argv) {
  const bool only_b = argc > 1 && argv[1][0] == 'B';
  if (!only_b) {
    run_rate<0>("v_and_b32");
    run_rate<1>("v_or_b32");
    run_rate<2>("v_xor_b32");
    run_rate<3>("v_add_u32");
    run_rate<4>("v_sub_u32");
    run_rate<5>("v_lshlrev_b32");
    run_rate<6>("v_lshrrev_b32");
    run_rate<7>("v_ashrrev_i32");
    run_rate<8>("v_mov_b32");
    run_rate<9>("v_mov_b32_dpp row_shr:1");
    run_rate<10>("v_mov_b32_dpp row_ror:1");
    run_rate<11>("v_mov_b32_dpp wave_shr:1");
    run_rate<12>("v_mov_b32_dpp quad_perm");
    run_rate<13>("v_add_u32_dpp row_shr:1");
    run_rate<14>("v_cndmask_b32");
    run_rate<15>("v_and_or_b32");
    run_rate<16>("v_or3_b32");
    run_rate<17>("v_lshl_or_b32");
    run_rate<18>("v_xnor_b32");
    run_rate<19>("v_not_b32");
    run_rate<20>("v_add_u16");
    run_rate<21>("v_sub_u16");
    run_rate<22>("v_mul_u32_u24");
    run_rate<23>("v_mad_u32_u24");
    run_rate<24>("v_perm_b32");
    run_rate<25>("v_alignbyte_b32");
    run_rate<26>("v_pk_sub_i16");
    run_rate<27>("v_dot4_i32_i8");
    run_rate<28>("v_bfi_b32");
    run_rate<29>("v_add_u32_sdwa WORD_1");
    run_rate<30>("v_max_u32");
    run_rate<31>("v_min_u32");
    run_rate<32>("v_xad_u32");
    run_rate<33>("v_add_lshl_u32");
    run_rate<34>("v_lshrrev_b16");
    run_rate<35>("v_pk_add_u16");
    run_rate<36>("v_sad_u8");
    run_rate<37>("v_subrev_u32");
    run_rate<39>("v_mul_lo_u32");
    run_rate<40>("v_dot4c_i32_i8 (VOP2)");
    run_rate<41>("v_dot4_u32_u8 (VOP3P)");
    run_rate<42>("v_dot2c_i32_i16 (VOP2)");
    run_rate<43>("v_dot8c_i32_i4 (VOP2)");
    run_rate<45>("v_fmac_f32 (VOP2)");
    {
      // a chain of dependent v_add_f64 on one wave (k1_flat_block's sequential sums): ns an addition
      double *dd;
      CK(hipMalloc(&dd, 64 * 8));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(f64_chain, dim3(1), dim3(64), 0, 0, dd, 1.0 + rep);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("F  dependent v_add_f64 chain: %d additions in %.1f us = %.2f ns an addition\n", 1 << 17, ms * 1e3, ms * 1e6 / (1 << 17));
      }
    }
  }
  int *d_in, *d_out;
  CK(hipMalloc(&d_in, 4096));
  CK(hipMemset(d_in, 1, 4096));
  CK(hipMalloc(&d_out, 4 * 1024 * 1024));
  // B1: 8 waves a CU = 2 per SIMD.  MFMA-only on both / VALU-only on both / one of each.  NV = VALU per MFMA slot:
  //     16 MFMAs ~ 16 x 20.4 = 326 cycles; 16 * NV alignbytes ~ 64 NV cycles: NV = 5 is the same time alone.
  run_co<0, 5>(8, d_in, d_out, "2 waves/SIMD, both MFMA-only (16 MFMA a round each)");
  run_co<1, 5>(8, d_in, d_out, "2 waves/SIMD, both VALU-only (80 v_alignbyte a round each)");
  run_co<2, 5>(8, d_in, d_out, "2 waves/SIMD, one MFMA-only + one VALU-only");
  run_co<0, 5>(4, d_in, d_out, "1 wave/SIMD MFMA-only");
  run_co<1, 5>(4, d_in, d_out, "1 wave/SIMD VALU-only");
  run_co<0, 5>(16, d_in, d_out, "4 waves/SIMD, all MFMA-only");
  run_co<1, 5>(16, d_in, d_out, "4 waves/SIMD, all VALU-only");
  run_co<2, 5>(16, d_in, d_out, "4 waves/SIMD, two MFMA-only + two VALU-only");
  // B2: one wave interleaving NV VALU after each MFMA
  run_co<3, 0>(4, d_in, d_out, "1 wave/SIMD, MFMA then NV VALU, interleaved");
  run_co<3, 1>(4, d_in, d_out, "1 wave/SIMD, MFMA then NV VALU, interleaved");
  run_co<3, 2>(4, d_in, d_out, "1 wave/SIMD, MFMA then NV VALU, interleaved");
  run_co<3, 3>(4, d_in, d_out, "1 wave/SIMD, MFMA then NV VALU, interleaved");
  run_co<3, 4>(4, d_in, d_out, "1 wave/SIMD, MFMA then NV VALU, interleaved");
  run_co<3, 5>(4, d_in, d_out, "1 wave/SIMD, MFMA then NV VALU, interleaved");
  run_co<3, 6>(4, d_in, d_out, "1 wave/SIMD, MFMA then NV VALU, interleaved");
  run_co<3, 8>(4, d_in, d_out, "1 wave/SIMD, MFMA then NV VALU, interleaved");
  run_co<3, 2>(16, d_in, d_out, "4 waves/SIMD, MFMA then NV VALU, interleaved");
  run_co<3, 4>(16, d_in, d_out, "4 waves/SIMD, MFMA then NV VALU, interleaved");
  run_co<3, 5>(16, d_in, d_out, "4 waves/SIMD, MFMA then NV VALU, interleaved");
  run_co<3, 8>(16, d_in, d_out, "4 waves/SIMD, MFMA then NV VALU, interleaved");
  if (!only_b) {
    run_lds<4, 0>("ds_write_b32 aligned");
    run_lds<4, 1>("ds_write_b32 +1 byte");
    run_lds<8, 0>("ds_write_b64 aligned");
    run_lds<8, 1>("ds_write_b64 +1 byte");
    run_lds<8, 3>("ds_write_b64 +3 bytes");
    run_lds<8, 4>("ds_write_b64 +4 bytes");
    run_lds<16, 0>("ds_write_b128 aligned");
    run_lds<16, 1>("ds_write_b128 +1 byte");
    run_lds<16, 4>("ds_write_b128 +4 bytes");
    run_lds<16, 8>("ds_write_b128 +8 bytes");
  }
  return 0;
}
