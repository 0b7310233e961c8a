// k1f.hip.h -- the certified fast path of the flat-block finder.
//
// The literal K1 (kernels.hip.h) reproduces FlatBlockFinder::run in the reference's f64
// operation order, one lane per block: 21 k dependent f64 operations per block.  Every
// feature it computes is, in exact arithmetic, a function of fourteen INTEGER sums over the
// block's 8-bit pixels p (v = p / 255, plane fit (yd c0 + xd c1) + c2 with c = M t):
//     gx = dx / 510 - c1 / 16,   gy = dy / 510 - c0 / 16,   dx = p(x+1) - p(x-1), dy likewise
//     sum gx^2 = DXX / 510^2 - c1 DX / 4080 + 900 c1^2 / 256,  ...  (see k1_certify)
// So:
//   k1_moments   the fourteen sums per block with v_dot4_u32_u8 / v_sad_u8 (exact, any order);
//   k1_certify   the features from the sums in plain f64 with a running bound on the distance
//                from their exact values (rounds 3 - 5: double-double arithmetic, five times the
//                instructions for the same open blocks), TOGETHER WITH a running bound on |what
//                the reference's rounded evaluation gives - this value| (standard forward error
//                analysis of its sequential sums, carried through every later operation).  A block whose four threshold tests
//                and whose f32 score are unambiguous within that bound gets them written;
//                any other block goes on a list;
//   k1_flat_features<.., true>   the literal kernel, for the listed blocks only.
// The result is bit for bit the literal kernel's wherever the bound holds; the bound carries a
// safety factor, and the tests compare every block of every test frame with the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pixel_helpers.hip.h"
#include "kernels.hip.h"

namespace g1s {

// ---------------------------------------------------------------------------------
// k1_moments<BPS>: lane = one row of a block (32 pixels as 8 packed dwords, loaded like the
// literal kernel does, edge replication included); 32 lanes = a block, two blocks per wave.
// grid = (ceil(nblocks / 8), batch), block = 256.
// ---------------------------------------------------------------------------------
template <int BPS>
__global__ __launch_bounds__(256) void k1_moments(const FrameTable ft, Geom g, int32_t *__restrict__ mom) {
  const int frame = blockIdx.y;
  const int tid = threadIdx.x, yi = tid & 31;
  const int blk = (int)blockIdx.x * 8 + (tid >> 5);
  const bool live = blk < g.nblocks;
  const FramePlanes fp = ft.f[frame];
  const int bx = live ? blk % g.nbw : 0, by = live ? blk / g.nbw : 0;
  const int ox = bx * kBlock, oy = by * kBlock;
  const bool fast = g.fast_rows && (ox + kBlock <= g.W);
  uint32_t pk[8];
  load_row32<BPS>(fp.src[0], fp.src_stride[0], g.src_shift, ox, min(oy + yi, g.H - 1), g.W, fast, pk);
  // rows above / below (same block: shuffles within 32 lanes)
  uint32_t pu[8], pd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // (DPP wave shifts: a move each, no LDS crossbar; the rows they get wrong -- the first and the last of a block, whose
    //  neighbour lane belongs to the other block of the wave -- are not interior rows and use neither)
    pu[k] = (uint32_t)__builtin_amdgcn_mov_dpp((int)pk[k], 0x138, 0xf, 0xf, true);  // wave_shr:1: the row above
    pd[k] = (uint32_t)__builtin_amdgcn_mov_dpp((int)pk[k], 0x130, 0xf, 0xf, true);  // wave_shl:1: the row below
  }
  int32_t s[14];
  row_moments(pk, pu, pd, yi, s);
  // the block's pixels inside the plane (get_block_mean's sum): the row's sum, but for the blocks on the plane's right and
  // bottom edges, whose replicated pixels do not count
  int32_t clip = s[kM_S0];
  if (!(fast && oy + kBlock <= g.H)) {
    clip = 0;
    if (oy + yi < g.H) {
      const int nvalid = min(g.W - ox, kBlock);
      uint32_t c = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int nv = min(max(nvalid - 4 * k, 0), 4);
        c = sad4(pk[k] & (nv == 4 ? 0xffffffffu : ((1u << (8 * nv)) - 1u)), c);
      }
      clip = (int32_t)c;
    }
  }
  // sum over the 32 rows of the block: the last four lanes of the block hold the sixteen totals between them
  int v16[16];
#pragma unroll
  for (int k = 0; k < 14; ++k) v16[k] = s[k];
  v16[kM_CLIP] = live ? clip : 0;
  v16[15] = 0;
  int x[4];
  half_sums_split16(v16, x);
  if (live && yi >= kBlock - 4) {
    int32_t *out = mom + ((size_t)frame * g.nblocks + blk) * kMomInts;
    const int c = yi & 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) out[4 * j + c] = x[j];
  }
}

// ---------------------------------------------------------------------------------
// k_dbg_coread (G1S_DBG_COREAD, a measurement aid; profiles/r06b_coread.txt): reads the luma source of the batch the way
// k1_moments does (lane = a block row of 64 bytes, 256-thread workgroups) with 12 registers and no LDS, so that one of its
// waves fits on a SIMD beside the four waves of an accumulation launch: what a pass over the planes costs UNDER that launch
// (three times what it costs behind it).  Writes nothing (the store's condition is never true; the loads cannot be dropped for it).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dbg_coread(const FrameTable ft, Geom g, int32_t *__restrict__ sink) {
  const int frame = blockIdx.y;
  const int tid = threadIdx.x, yi = tid & 31;
  const int blk = (int)blockIdx.x * 8 + (tid >> 5);
  if (blk >= g.nblocks) return;
  const int bx = blk % g.nbw, by = blk / g.nbw;
  const int ox = bx * kBlock, y = min(by * kBlock + yi, g.H - 1);
  if (ox + kBlock > g.W) return;
  const FramePlanes fp = ft.f[frame];
  gptr_u4 p = (gptr_u4)(as_global(fp.src[0]) + (size_t)y * fp.src_stride[0] + (size_t)ox * g.src_bps);
  uint4 a = gload4(p), b = gload4(p + 1);
  uint32_t x = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
  if (g.src_bps == 2) {
    a = gload4(p + 2), b = gload4(p + 3);
    x ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
  }
  if (x == 0x9e3779b9u && sink[0] == 0x12345678) sink[1] = (int32_t)x;
}

// ---- a value the way the reference computes it, known up to a bound ----
// v: this kernel's f64 evaluation; e: bound on |reference's rounded evaluation - v|.
struct VE {
  double v, e;
};
constexpr double kU = 1.1102230246251565e-16;  // 2^-53
__device__ __forceinline__ VE ve_add(VE a, VE b) {
  const double v = a.v + b.v;
  return VE{v, a.e + b.e + 2.0 * kU * fabs(v)};
}
__device__ __forceinline__ VE ve_sub(VE a, VE b) {
  const double v = a.v - b.v;
  return VE{v, a.e + b.e + 2.0 * kU * fabs(v)};
}
__device__ __forceinline__ VE ve_mul(VE a, VE b) {
  const double v = a.v * b.v;
  return VE{v, fabs(a.v) * b.e + fabs(b.v) * a.e + a.e * b.e + 2.0 * kU * fabs(v)};
}
__device__ __forceinline__ VE ve_mul_c(double c, VE a) {  // exact constant
  const double v = c * a.v;
  return VE{v, fabs(c) * a.e + 2.0 * kU * fabs(v)};
}
__device__ __forceinline__ VE ve_div_c(VE a, double c) {
  // (a multiplication by the rounded reciprocal: within 1.5 ulp of the quotient the reference rounds once -- 4 u instead of 2 u
  //  in the bound, a dozen instructions less than the division)
  const double rc = 1.0 / c;  // (c is a constant of the evaluation: folded)
  const double v = a.v * rc;
  return VE{v, a.e * fabs(rc) * (1.0 + 4.0 * kU) + 4.0 * kU * fabs(v)};
}
__device__ __forceinline__ VE ve_add_c(VE a, double c) {
  const double v = a.v + c;
  return VE{v, a.e + 2.0 * kU * fabs(v)};
}
// 1: certainly a < K; 0: certainly not; -1: cannot tell
__device__ __forceinline__ int ve_lt(VE a, double K) { return a.v + a.e < K ? 1 : (a.v - a.e >= K ? 0 : -1); }
__device__ __forceinline__ int ve_gt(VE a, double K) { return a.v - a.e > K ? 1 : (a.v + a.e <= K ? 0 : -1); }

// ---- a value of the EXACT evaluation (the real-number function of the integer sums), computed in plain f64: |v - exact| <= e ----
// (standard model: a rounded operation is within u of its result; 2 u taken; the bound's own roundings are covered where the
//  bounds are used, k1_certify's kInfl)
struct PE {
  double v, e;
};
__device__ __forceinline__ PE pe_add(PE a, PE b) {
  const double v = a.v + b.v;
  return PE{v, __builtin_fma(2.0 * kU, fabs(v), a.e + b.e)};
}
__device__ __forceinline__ PE pe_sub(PE a, PE b) {
  const double v = a.v - b.v;
  return PE{v, __builtin_fma(2.0 * kU, fabs(v), a.e + b.e)};
}
__device__ __forceinline__ PE pe_mul(PE a, PE b) {
  const double v = a.v * b.v;
  return PE{v, __builtin_fma(2.0 * kU, fabs(v), __builtin_fma(fabs(a.v), b.e, __builtin_fma(fabs(b.v), a.e, a.e * b.e)))};
}
__device__ __forceinline__ PE pe_mul_k(PE a, double k) {  // k: an exact factor (a constant of the evaluation, an integer sum)
  const double v = a.v * k;
  return PE{v, __builtin_fma(2.0 * kU, fabs(v), fabs(k) * a.e)};
}
__device__ __forceinline__ PE pe_mul_r(PE a, double rc) {  // rc = fl(1 / c) > 0 for a / c: the factor itself is within u of 1 / c
  const double v = a.v * rc;
  return PE{v, __builtin_fma(4.0 * kU, fabs(v), rc * a.e)};
}

// ---- square roots and reciprocals that only feed BOUNDS: the f32 instruction (1 ulp) widened to the safe side, four
// instructions where the IEEE f64 forms are fifteen to twenty.  Arguments below the f32 range come out as 0 (the root's floor)
// or as 1.1e-19 (its ceiling: sqrt of the smallest normal f32); a reciprocal of such an argument is infinite -- the block then
// stays open and goes to the literal kernel. ----
__device__ __forceinline__ double sqrt_up(double z) {  // >= sqrt(z), z >= 0
  return __builtin_fma((double)__builtin_amdgcn_sqrtf((float)z), 1.0 + 1e-6, 1.1e-19);
}
__device__ __forceinline__ double sqrt_down(double z) {  // <= sqrt(z), z >= 0
  return (double)__builtin_amdgcn_sqrtf((float)z) * (1.0 - 1e-6);
}
__device__ __forceinline__ double rcp_up(double d) {  // >= 1 / d, d > 0
  return (double)__builtin_amdgcn_rcpf((float)d) * (1.0 + 1e-6);
}

struct CertifyLists {
  uint32_t *list;   // [batch][nblocks] blocks left to the literal kernel -- or, `global`, one sequence of frame * nblocks + block
  uint32_t *count;  // [batch] -- global: count[0] the sequence's length
  int global;
};

// ---------------------------------------------------------------------------------
// k1_certify: one thread per block.  grid = (ceil(nblocks / 256), batch), block = 256.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k1_certify(Geom g, FlatConsts fc, const int32_t *__restrict__ mom,
                                                  uint8_t *__restrict__ records, uint8_t *__restrict__ flags,
                                                  CertifyLists cl, int force_literal) {
  const int frame = blockIdx.y;
  const int blk = (int)blockIdx.x * 256 + (int)threadIdx.x;
  bool certain = false;
  float score_out = 0.0f;
  uint8_t flag_out = 0;
  if (blk < g.nblocks && !force_literal) {
    const int32_t *m = mom + ((size_t)frame * g.nblocks + blk) * kMomInts;
    const double S0 = m[kM_S0], SX = (double)m[kM_SXU] - 16.0 * m[kM_S0], SY = (double)m[kM_SYU] - 16.0 * m[kM_S0];
    const double SAX = m[kM_SAX], SAY = m[kM_SAY];
    const double I0 = m[kM_I0], IX = (double)m[kM_IXU] - 16.0 * m[kM_I0], IY = (double)m[kM_IYU] - 16.0 * m[kM_I0];
    const double IPP = m[kM_IPP], DXX = m[kM_DXX], DYY = m[kM_DYY], DXY = m[kM_DXY], DX = m[kM_DX], DY = m[kM_DY];
    const double N = 900.0;
    // reference: 1024 products v*yd (v itself rounded) summed sequentially
    constexpr double kEt01 = 1030.0 * kU / 4080.0 * (1.0 + 1e-12), kEt2 = 1026.0 * kU / 255.0 * (1.0 + 1e-12);
    const double Et0 = kEt01 * SAY, Et1 = kEt01 * SAX, Et2 = kEt2 * S0;
    // The five interior sums (and the fit's t, c they are made of) as functions of the integer sums, in plain f64 with a running
    // bound on |value - exact| (PE).  Rounds 3 - 5 evaluated them in double-double arithmetic (error 2^-100; 1 100 of the kernel's
    // 1 700 instructions): the reference's OWN roundings (the 905 u / 1030 u terms below) are a hundred times a plain evaluation's,
    // so the plain one with its bound added leaves the same blocks open (147 -> 148 of 81 720 on the stress frames, none wrong)
    // at 33 -> 22 us a 64-frame 4K launch (profiles/r05r_certify.txt).
    // v[]: sGxx, sGyy, sGxy, sR, sR2; x[]: |v - exact| bounds; cv / cx: c0 .. c2; tv / tx: t0 .. t2.
    double v[5], x[5], cv[3], cx[3], tv[3], tx[3];
    {
      // plane fit: t = (sum v yd, sum v xd, sum v), c = M t
      const PE t0 = pe_mul_r(PE{SY, 0.0}, 1.0 / 4080.0), t1 = pe_mul_r(PE{SX, 0.0}, 1.0 / 4080.0), t2 = pe_mul_r(PE{S0, 0.0}, 1.0 / 255.0);
      PE c[3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
        c[i] = pe_add(pe_add(pe_mul_k(t0, fc.ata_inv[3 * i]), pe_mul_k(t1, fc.ata_inv[3 * i + 1])), pe_mul_k(t2, fc.ata_inv[3 * i + 2]));
      const PE c0 = c[0], c1 = c[1], c2 = c[2];
      const PE c00 = pe_mul(c0, c0), c11 = pe_mul(c1, c1), c01 = pe_mul(c0, c1), c0p1 = pe_add(c0, c1);
      // sum gx^2 = DXX/510^2 - c1 DX/4080 + N c1^2/256
      const PE sGxx = pe_add(pe_sub(pe_mul_r(PE{DXX, 0.0}, 1.0 / 260100.0), pe_mul_r(pe_mul_k(c1, DX), 1.0 / 4080.0)), pe_mul_k(c11, N / 256.0));
      const PE sGyy = pe_add(pe_sub(pe_mul_r(PE{DYY, 0.0}, 1.0 / 260100.0), pe_mul_r(pe_mul_k(c0, DY), 1.0 / 4080.0)), pe_mul_k(c00, N / 256.0));
      // sum gx gy = DXY/510^2 - c0 DX/8160 - c1 DY/8160 + N c0 c1/256
      const PE sGxy = pe_add(pe_sub(pe_mul_r(PE{DXY, 0.0}, 1.0 / 260100.0), pe_mul_r(pe_mul_k(c0, DX), 1.0 / 8160.0)),
                             pe_sub(pe_mul_k(c01, N / 256.0), pe_mul_r(pe_mul_k(c1, DY), 1.0 / 8160.0)));
      // sum r = I0/255 - c0 sumY/16 - c1 sumX/16 - N c2,  sumX = sumY = -450 over the interior
      const PE sR = pe_sub(pe_add(pe_mul_r(PE{I0, 0.0}, 1.0 / 255.0), pe_mul_k(c0p1, 450.0 / 16.0)), pe_mul_k(c2, N));
      // sum r^2 = IPP/255^2 - 2 [c0 IY/4080 + c1 IX/4080 + c2 I0/255] + sum fit^2
      const PE cross = pe_add(pe_add(pe_mul_r(pe_mul_k(c0, IY), 1.0 / 4080.0), pe_mul_r(pe_mul_k(c1, IX), 1.0 / 4080.0)),
                              pe_mul_r(pe_mul_k(c2, I0), 1.0 / 255.0));
      // sum fit^2 = (c0^2 + c1^2) 67650/256 + N c2^2 + 2 c0 c1 225/256 + 2 (c0 + c1) c2 (-450)/16
      const PE fit2 = pe_add(pe_add(pe_mul_k(pe_add(c00, c11), 67650.0 / 256.0), pe_mul_k(pe_mul(c2, c2), N)),
                             pe_sub(pe_mul_k(c01, 450.0 / 256.0), pe_mul_k(pe_mul(c0p1, c2), 900.0 / 16.0)));
      const PE sR2 = pe_add(pe_sub(pe_mul_r(PE{IPP, 0.0}, 1.0 / 65025.0), pe_mul_k(cross, 2.0)), fit2);
      // (the bounds' own arithmetic rounds too -- a few u of each, relative: taken in once, generously)
      constexpr double kInfl = 1.0 + 1e-7;
      v[0] = sGxx.v, v[1] = sGyy.v, v[2] = sGxy.v, v[3] = sR.v, v[4] = sR2.v;
      x[0] = sGxx.e * kInfl, x[1] = sGyy.e * kInfl, x[2] = sGxy.e * kInfl, x[3] = sR.e * kInfl, x[4] = sR2.e * kInfl;
      cv[0] = c0.v, cv[1] = c1.v, cv[2] = c2.v, cx[0] = c0.e * kInfl, cx[1] = c1.e * kInfl, cx[2] = c2.e * kInfl;
      tv[0] = t0.v, tv[1] = t1.v, tv[2] = t2.v, tx[0] = t0.e * kInfl, tx[1] = t1.e * kInfl, tx[2] = t2.e * kInfl;
    }
    {
      // ---- how far the reference's c can be from the exact c (its t are sequential sums of rounded products) ----
      double Ec[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double m0 = fc.ata_inv[3 * i], m1 = fc.ata_inv[3 * i + 1], m2 = fc.ata_inv[3 * i + 2];
        Ec[i] = fabs(m0) * (Et0 + 4.0 * kU * (fabs(tv[0]) + tx[0])) + fabs(m1) * (Et1 + 4.0 * kU * (fabs(tv[1]) + tx[1])) +
                fabs(m2) * (Et2 + 4.0 * kU * (fabs(tv[2]) + tx[2]));
      }
      const double F = (fabs(cv[0]) + cx[0]) + (fabs(cv[1]) + cx[1]) + (fabs(cv[2]) + cx[2]) + Ec[0] + Ec[1] + Ec[2];
      const double Eg0 = kU * (3.0 + 5.0 * F);
      const double Egx = Ec[1] / 16.0 + Eg0, Egy = Ec[0] / 16.0 + Eg0;
      const double Er = kU * (2.0 + 5.0 * F) + Ec[0] + Ec[1] + Ec[2];
      // (the sums where the bounds below want the EXACT ones: from above, by the plain evaluation's own bound)
      const double gxx = fmax(v[0] + x[0], 0.0), gyy = fmax(v[1] + x[1], 0.0), r2 = fmax(v[4] + x[4], 0.0);
      // ---- how far the reference's rounded sequential sums can be from these ----
      const double kSafety = 4.0;
      const double sxx = sqrt_up(N * gxx), syy = sqrt_up(N * gyy), sr2 = sqrt_up(N * r2);  // (factors of bounds)
      VE Gxx{v[0], kSafety * (905.0 * kU * gxx + 2.0 * Egx * sxx + N * Egx * Egx) + x[0]};
      VE Gyy{v[1], kSafety * (905.0 * kU * gyy + 2.0 * Egy * syy + N * Egy * Egy) + x[1]};
      VE Gxy{v[2], kSafety * (905.0 * kU * 0.5 * (gxx + gyy) + Egy * sxx + Egx * syy + N * Egx * Egy) + x[2]};
      VE mean{v[3], kSafety * (905.0 * kU * sr2 + N * (kU * (2.0 + 5.0 * F) + Ec[2] + (Ec[0] + Ec[1]) / 32.0)) + x[3]};
      VE var{v[4], kSafety * (905.0 * kU * r2 + 2.0 * Er * sr2 + N * Er * Er) + x[4]};
      // ---- the rest of the reference's evaluation, bound carried along ----
      mean = ve_div_c(mean, N);
      Gxx = ve_div_c(Gxx, N);
      Gxy = ve_div_c(Gxy, N);
      Gyy = ve_div_c(Gyy, N);
      var = ve_sub(ve_div_c(var, N), ve_mul(mean, mean));
      const VE trace = ve_add(Gxx, Gyy);
      const VE det = ve_sub(ve_mul(Gxx, Gyy), ve_mul(Gxy, Gxy));
      VE disc = ve_sub(ve_mul(trace, trace), ve_mul_c(4.0, det));
      if (!(disc.v > 0.0)) disc.v = 0.0;  // (the reference clamps too: |max(a,0) - max(b,0)| <= |a - b|)
      VE sq;
      sq.v = sqrt(disc.v);
      {
        // |sqrt(a) - sqrt(b)| <= |a - b| / (2 sqrt(min(a, b))) and <= sqrt(|a - b|): the first where the bracket's lower end is
        // well above zero (there it is the smaller of the two), the second otherwise
        const double lo = disc.v - disc.e;
        const bool big = lo > disc.e;
        sq.e = (big ? disc.e * rcp_up(2.0 * sqrt_down(lo)) : sqrt_up(disc.e)) + 2.0 * kU * sq.v;
      }
      const VE e1 = ve_div_c(ve_add(trace, sq), 2.0);
      const VE e2 = ve_div_c(ve_sub(trace, sq), 2.0);
      const VE norm = e1;
      VE den = e2;
      if (!(den.v > 1e-6)) den.v = 1e-6;  // max(e2, 1e-6): 1-Lipschitz
      VE ratio{e1.v / den.v, 0.0};
      const double den_lo = den.v - den.e;
      const bool ratio_ok = den_lo > 0.0;
      ratio.e = ratio_ok ? (e1.e + fabs(ratio.v) * den.e) * rcp_up(den_lo) * (1.0 + 8.0 * kU) + 2.0 * kU * fabs(ratio.v) : 1e300;
      const double kTrace = 0.15 / 1024.0, kRatio = 1.25, kNorm = 0.08 / 1024.0, kVar = 0.005 / 1024.0;
      const int tr_lt = ve_lt(trace, kTrace), ra_lt = ve_lt(ratio, kRatio), no_lt = ve_lt(norm, kNorm), va_gt = ve_gt(var, kVar);
      // is_flat = all four; certain if every test is, or if one is certainly false
      int flat;
      if (tr_lt == 0 || ra_lt == 0 || no_lt == 0 || va_gt == 0) flat = 0;
      else if (tr_lt == 1 && ra_lt == 1 && no_lt == 1 && va_gt == 1) flat = 1;
      else flat = -1;
      // score
      VE sw = ve_mul_c(-6682.0, var);
      sw = ve_add(sw, ve_mul_c(-0.2056, ratio));
      sw = ve_add(sw, ve_mul_c(13087.0, trace));
      sw = ve_add(sw, ve_mul_c(-12434.0, norm));
      sw = ve_add_c(sw, 2.5694);
      // clamp, then s = 1 / (1 + exp(-sw)), monotone in sw: the reference's sw lies in [sw.v - e, sw.v + e],
      // so its s lies between the images of the ends (widened by the roundings of exp, + and /)
      auto clamp_sw = [](double z) { return z < -25.0 ? -25.0 : (z > 100.0 ? 100.0 : z); };
      auto sigmoid = [](double z) { return 1.0 / (1.0 + exp(-z)); };
      const double sv = sigmoid(clamp_sw(sw.v));
      // The images of the interval's ends without evaluating them (two exponentials, two divisions): x -> sigmoid(clamp(x)) has
      // slope s (1 - s) <= 1 / 4 and |second derivative| <= 1 / (6 sqrt 3) < 0.1, so over [sw.v - e, sw.v + e] it stays within
      // e (s (1 - s) + 0.1 e) of sv; the few ulps of sv's own evaluation are inside the 16 u factors below.
      // (the interval as far as the clamp lets it through: a block far outside [-25, 100] has none left)
      const double x_c = clamp_sw(sw.v);
      const double e_c = fmax(clamp_sw(sw.v + sw.e) - x_c, x_c - clamp_sw(sw.v - sw.e)) * (1.0 + 4.0 * kU);
      const double s_dev = e_c * (sv * (1.0 - sv) + 0.1 * e_c + 8.0 * kU) * (1.0 + 8.0 * kU);
      const double s_lo = sv - s_dev, s_hi = sv + s_dev;
      const float f_lo = (float)(s_lo * (1.0 - 16.0 * kU)), f_hi = (float)(s_hi * (1.0 + 16.0 * kU));
      const bool score_ok = ratio_ok && (f_lo == f_hi) && isfinite(sw.e);
      if (flat >= 0 && va_gt >= 0 && (va_gt == 0 || score_ok)) {
        certain = true;
        flag_out = flat ? 255 : 0;
        // (the float both ends of the bracket round to -- sv lies inside it, and so does the reference's value)
        score_out = va_gt ? f_lo : 0.0f;
      }
    }
  }
  if (blk < g.nblocks) {
    uint8_t *rec = records + (size_t)frame * g.rec_size;
    // the record's luma_sum (get_block_mean as an exact sum) comes from the finder's moments: the accumulation launches do
    // not form it (4 additions a row word of the luma launch)
    reinterpret_cast<uint32_t *>(rec + g.off_luma_sum)[blk] = (uint32_t)mom[((size_t)frame * g.nblocks + blk) * kMomInts + kM_CLIP];
    if (certain) {
      reinterpret_cast<float *>(rec + g.off_scores)[blk] = score_out;
      flags[(size_t)frame * g.nblocks + blk] = flag_out;
    }
  }
  // the rest: compacted for the literal kernel (one atomic per wave)
  const bool todo = blk < g.nblocks && !certain;
  const unsigned long long bal = __ballot(todo);
  if (bal != 0) {
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&cl.count[cl.global ? 0 : frame], (uint32_t)__popcll(bal));
    base = __shfl(base, 0, 64);
    const uint32_t at = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    if (todo) {
      if (cl.global) cl.list[at] = (uint32_t)frame * (uint32_t)g.nblocks + (uint32_t)blk;
      else cl.list[(size_t)frame * g.nblocks + at] = (uint32_t)blk;
    }
  }
}

}  // namespace g1s
