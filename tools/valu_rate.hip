// valu_rate.hip -- instruction-throughput microbenchmark for the candidate
// inner-loop instructions of the AR accumulation kernel (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short short2_ __attribute__((ext_vector_type(2)));
typedef _Float16 half2_ __attribute__((ext_vector_type(2)));
typedef float float2_ __attribute__((ext_vector_type(2)));

#define NACC 32
#define ITERS 2048

template <int OP>
__global__ __launch_bounds__(256) void rate(int *out, int a0, int b0) {
  int acc[NACC];
  float facc[NACC];
  int a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
#pragma unroll
  for (int i = 0; i < NACC; ++i) { acc[i] = i; facc[i] = (float)i; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (OP == 0) acc[i] = __builtin_amdgcn_sdot4(a, b + i, acc[i], false);
      if (OP == 1) acc[i] = __builtin_amdgcn_udot4(a, b + i, acc[i], false);
      if (OP == 2) { short2_ x = __builtin_bit_cast(short2_, a), y = __builtin_bit_cast(short2_, b + i); acc[i] = __builtin_amdgcn_sdot2(x, y, acc[i], false); }
      if (OP == 3) acc[i] = ((a << 8) >> 8) * (((b + i) << 8) >> 8) + acc[i];  // 24-bit operands -> v_mad_i32_i24
      if (OP == 4) acc[i] = a * (b + i) + acc[i];                      // 32-bit mul + add
      if (OP == 5) facc[i] = __builtin_fmaf((float)a, facc[i], (float)b);
      if (OP == 6) { half2_ x = __builtin_bit_cast(half2_, a), y = __builtin_bit_cast(half2_, b + i); facc[i] = __builtin_amdgcn_fdot2(x, y, facc[i], false); }
      if (OP == 7) acc[i] = __builtin_amdgcn_alignbyte(a, acc[i], 1);
      if (OP == 8) acc[i] = (acc[i] & a) + b;
      if (OP == 9) acc[i] = __builtin_amdgcn_sdot8(a, b + i, acc[i], false);
      if (OP == 10) acc[i] = __builtin_amdgcn_perm(a, acc[i], 0x04030201u);
      if (OP == 11) acc[i] = __builtin_amdgcn_alignbit(a, acc[i], 8);
      if (OP == 12) acc[i] = (int)__builtin_amdgcn_sad_u8((unsigned)a, (unsigned)(b + i), (unsigned)acc[i]);
      if (OP == 13) acc[i] = (acc[i] >> 8) | (a << 24);
      if (OP == 14) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
      if (OP == 15) asm volatile("v_pk_sub_i16 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
      if (OP == 16) asm volatile("v_pk_max_i16 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
      if (OP == 17) asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (OP == 18) asm volatile("v_pk_lshrrev_b16 %0, 3, %0" : "+v"(acc[i]));
      if (OP == 19) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (OP == 20) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (OP == 21) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (OP == 22) asm volatile("v_lshl_add_u32 %0, %1, 1, %0" : "+v"(acc[i]) : "v"(a));
      if (OP == 23) asm volatile("v_max_i32 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
      if (OP == 24) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
      if (OP == 25) asm volatile("v_pk_mul_lo_u16 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
      if (OP == 26) asm volatile("v_and_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
      if (OP == 27) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(acc[i]));
      if (OP == 28) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(acc[i]));
      if (OP == 29) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (OP == 30) asm volatile("v_cmp_lt_i32 vcc, %1, %0\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(acc[i]) : "v"(a) : "vcc");
      if (OP == 31) asm volatile("v_pk_add_i16 %0, %1, %0 neg_lo:[1,0] neg_hi:[1,0]" : "+v"(acc[i]) : "v"(a));
      if (OP == 32) asm volatile("v_pk_ashrrev_i16 %0, 15, %0" : "+v"(acc[i]));
    }
    a += it;
  }
  int s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i] + (int)facc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, int macs_per_instr) {
  int *d;
  const int blocks = 256 * 8;
  hipMalloc(&d, blocks * 256 * sizeof(int));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  rate<OP><<<blocks, 256>>>(d, 3, 5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  rate<OP><<<blocks, 256>>>(d, 3, 5);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)blocks * 256 * ITERS * NACC;   // lane-instructions
  const double gips = instr / (ms * 1e-3) / 1e12;
  // 256 CU * 4 SIMD * 32 lanes * 2.4 GHz = 78.6 T lane-instr/s at 1 per lane per clk
  printf("%-22s %8.3f ms  %7.2f T lane-instr/s  (%.2f of 78.6)  %7.1f T MAC/s\n", name, ms, gips, gips / 78.6, gips * macs_per_instr);
  hipFree(d);
}
int main() {
  run<0>("v_dot4c_i32_i8", 4);
  run<1>("v_dot4_u32_u8", 4);
  run<2>("v_dot2c_i32_i16", 2);
  run<3>("v_mad_i32_i24", 1);
  run<4>("v_mul_lo_u32+add", 1);
  run<5>("v_fma_f32", 1);
  run<6>("v_dot2_f32_f16", 2);
  run<7>("v_alignbyte_b32", 0);
  run<8>("v_and+v_add", 0);
  run<9>("v_dot8_i32_i4", 8);
  run<10>("v_perm_b32", 0);
  run<11>("v_alignbit_b32", 0);
  run<12>("v_sad_u8", 0);
  run<13>("v_lshr+v_lshl_or (2)", 0);
  run<14>("v_pk_add_u16", 0);
  run<15>("v_pk_sub_i16", 0);
  run<16>("v_pk_max_i16", 0);
  run<17>("v_pk_mad_u16", 0);
  run<18>("v_pk_lshrrev_b16", 0);
  run<19>("v_dot2_u32_u16", 2);
  run<20>("v_sad_u16", 0);
  run<21>("v_add3_u32", 0);
  run<22>("v_lshl_add_u32", 0);
  run<23>("v_max_i32", 0);
  run<24>("v_sub_u32", 0);
  run<25>("v_pk_mul_lo_u16", 0);
  run<26>("v_and_b32", 0);
  run<27>("v_lshrrev_b32", 0);
  run<28>("v_bfe_u32", 0);
  run<29>("v_sad_u32", 0);
  run<30>("v_cmp+v_addc (2)", 0);
  run<31>("v_pk_add_i16 neg", 0);
  run<32>("v_pk_ashrrev_i16", 0);
  return 0;
}
