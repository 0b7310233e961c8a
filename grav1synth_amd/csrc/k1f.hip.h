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
//   k1_certify   the features from the sums in double-double arithmetic (no cancellation),
//                TOGETHER WITH a running bound on |what the reference's rounded evaluation
//                gives - this value| (standard forward error analysis of its sequential sums,
//                carried through every later operation).  A block whose four threshold tests
//                and whose f32 score are unambiguous within that bound gets them written;
//                any other block goes on a list;
//   k1_flat_features<.., true>   the literal kernel, for the listed blocks only.
// The result is bit for bit the literal kernel's wherever the bound holds; the bound carries a
// safety factor, and the tests compare every block of every test frame with the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k0.hip.h"
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

// ---- double-double arithmetic (error-free transformations; FMA is explicit here) ----
struct dd {
  double hi, lo;
};
__device__ __forceinline__ dd dd_from(double a) { return dd{a, 0.0}; }
__device__ __forceinline__ dd two_sum(double a, double b) {
  const double s = a + b, bb = s - a;
  return dd{s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd two_prod(double a, double b) {
  const double p = a * b;
  return dd{p, __builtin_fma(a, b, -p)};
}
__device__ __forceinline__ dd dd_add(dd a, dd b) {
  dd s = two_sum(a.hi, b.hi);
  s.lo += a.lo + b.lo;
  return two_sum(s.hi, s.lo);
}
__device__ __forceinline__ dd dd_neg(dd a) { return dd{-a.hi, -a.lo}; }
__device__ __forceinline__ dd dd_mul(dd a, dd b) {
  dd p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd dd_mul_d(dd a, double b) { return dd_mul(a, dd_from(b)); }
// a / b for a double divisor, to ~2^-100
__device__ __forceinline__ dd dd_div_d(dd a, double b) {
  const double q1 = a.hi / b;
  const dd p = two_prod(q1, b);
  const double r = ((a.hi - p.hi) - p.lo) + a.lo;
  const double q2 = r / b;
  return two_sum(q1, q2);
}

// a / c for the constants of the evaluation, as a multiplication by the double-double reciprocal (hi + lo = 1 / c to 2^-108;
// the product to ~2^-103): the two f64 divisions of dd_div_d were a third of k1_certify's instructions
constexpr dd kRcp4080{0.00024509803921568627, 3.4014185803466805e-21};
constexpr dd kRcp255{0.00392156862745098, 5.442269728554689e-20};
constexpr dd kRcp260100{3.844675124951942e-06, -4.216472044548166e-22};
constexpr dd kRcp8160{0.00012254901960784314, 1.7007092901733403e-21};
constexpr dd kRcp65025{1.5378700499807768e-05, -1.6865888178192663e-21};

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
    // ---- plane fit: t = (sum v yd, sum v xd, sum v), c = M t ----
    const dd t0 = dd_mul(dd_from(SY), kRcp4080), t1 = dd_mul(dd_from(SX), kRcp4080), t2 = dd_mul(dd_from(S0), kRcp255);
    // reference: 1024 products v*yd (v itself rounded) summed sequentially
    const double Et0 = 1030.0 * kU * SAY / 4080.0, Et1 = 1030.0 * kU * SAX / 4080.0, Et2 = 1026.0 * kU * S0 / 255.0;
    dd c[3];
    double Ec[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double m0 = fc.ata_inv[3 * i], m1 = fc.ata_inv[3 * i + 1], m2 = fc.ata_inv[3 * i + 2];
      c[i] = dd_add(dd_add(dd_mul_d(t0, m0), dd_mul_d(t1, m1)), dd_mul_d(t2, m2));
      Ec[i] = fabs(m0) * (Et0 + 4.0 * kU * fabs(t0.hi)) + fabs(m1) * (Et1 + 4.0 * kU * fabs(t1.hi)) +
              fabs(m2) * (Et2 + 4.0 * kU * fabs(t2.hi));
    }
    const double F = fabs(c[0].hi) + fabs(c[1].hi) + fabs(c[2].hi) + Ec[0] + Ec[1] + Ec[2];
    const double Eg0 = kU * (3.0 + 5.0 * F);
    const double Egx = Ec[1] / 16.0 + Eg0, Egy = Ec[0] / 16.0 + Eg0;
    const double Er = kU * (2.0 + 5.0 * F) + Ec[0] + Ec[1] + Ec[2];
    // ---- the five interior sums, exactly ----
    const dd c0 = c[0], c1 = c[1], c2 = c[2];
    const dd c00 = dd_mul(c0, c0), c11 = dd_mul(c1, c1), c01 = dd_mul(c0, c1);
    // sum gx^2 = DXX/510^2 - c1 DX/4080 + N c1^2/256
    const dd sGxx = dd_add(dd_add(dd_mul(dd_from(DXX), kRcp260100), dd_neg(dd_mul(dd_mul_d(c1, DX), kRcp4080))),
                           dd_mul_d(c11, N / 256.0));
    const dd sGyy = dd_add(dd_add(dd_mul(dd_from(DYY), kRcp260100), dd_neg(dd_mul(dd_mul_d(c0, DY), kRcp4080))),
                           dd_mul_d(c00, N / 256.0));
    // sum gx gy = DXY/510^2 - c0 DX/8160 - c1 DY/8160 + N c0 c1/256
    const dd sGxy = dd_add(dd_add(dd_mul(dd_from(DXY), kRcp260100), dd_neg(dd_mul(dd_mul_d(c0, DX), kRcp8160))),
                           dd_add(dd_neg(dd_mul(dd_mul_d(c1, DY), kRcp8160)), dd_mul_d(c01, N / 256.0)));
    // sum r = I0/255 - c0 sumY/16 - c1 sumX/16 - N c2,  sumX = sumY = -450 over the interior
    const dd sR = dd_add(dd_add(dd_mul(dd_from(I0), kRcp255), dd_mul_d(dd_add(c0, c1), 450.0 / 16.0)), dd_neg(dd_mul_d(c2, N)));
    // sum r^2 = IPP/255^2 - 2 [c0 IY/4080 + c1 IX/4080 + c2 I0/255] + sum fit^2
    const dd cross = dd_add(dd_add(dd_mul(dd_mul_d(c0, IY), kRcp4080), dd_mul(dd_mul_d(c1, IX), kRcp4080)),
                            dd_mul(dd_mul_d(c2, I0), kRcp255));
    // sum fit^2 = (c0^2 + c1^2) 67650/256 + N c2^2 + 2 c0 c1 225/256 + 2 (c0 + c1) c2 (-450)/16
    const dd fit2 = dd_add(dd_add(dd_mul_d(dd_add(c00, c11), 67650.0 / 256.0), dd_mul_d(dd_mul(c2, c2), N)),
                           dd_add(dd_mul_d(c01, 450.0 / 256.0), dd_neg(dd_mul_d(dd_mul(dd_add(c0, c1), c2), 900.0 / 16.0))));
    const dd sR2 = dd_add(dd_add(dd_mul(dd_from(IPP), kRcp65025), dd_neg(dd_mul_d(cross, 2.0))), fit2);
    const double gxx = fmax(sGxx.hi, 0.0), gyy = fmax(sGyy.hi, 0.0), r2 = fmax(sR2.hi, 0.0);
    // ---- how far the reference's rounded sequential sums can be from these ----
    const double kSafety = 4.0;
    const double sxx = sqrt(N * gxx), syy = sqrt(N * gyy), sr2 = sqrt(N * r2);
    VE Gxx{sGxx.hi, kSafety * (905.0 * kU * gxx + 2.0 * Egx * sxx + N * Egx * Egx)};
    VE Gyy{sGyy.hi, kSafety * (905.0 * kU * gyy + 2.0 * Egy * syy + N * Egy * Egy)};
    VE Gxy{sGxy.hi, kSafety * (905.0 * kU * 0.5 * (gxx + gyy) + Egy * sxx + Egx * syy + N * Egx * Egy)};
    VE mean{sR.hi, kSafety * (905.0 * kU * sr2 + N * (kU * (2.0 + 5.0 * F) + Ec[2] + (Ec[0] + Ec[1]) / 32.0))};
    VE var{sR2.hi, kSafety * (905.0 * kU * r2 + 2.0 * Er * sr2 + N * Er * Er)};
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
      const double lo = disc.v - disc.e;
      sq.e = (lo > 0.0 ? disc.e / (2.0 * sqrt(lo)) : sqrt(disc.e + disc.v)) + 2.0 * kU * sq.v;
      sq.e = fmin(sq.e, sqrt(disc.e) + 2.0 * kU * sq.v + (lo > 0.0 ? 0.0 : 0.0));
    }
    const VE e1 = ve_div_c(ve_add(trace, sq), 2.0);
    const VE e2 = ve_div_c(ve_sub(trace, sq), 2.0);
    const VE norm = e1;
    VE den = e2;
    if (!(den.v > 1e-6)) den.v = 1e-6;  // max(e2, 1e-6): 1-Lipschitz
    VE ratio{e1.v / den.v, 0.0};
    const double den_lo = den.v - den.e;
    const bool ratio_ok = den_lo > 0.0;
    ratio.e = ratio_ok ? (e1.e + fabs(ratio.v) * den.e) / den_lo + 2.0 * kU * fabs(ratio.v) : 1e300;
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
    auto clamp_sw = [](double x) { return x < -25.0 ? -25.0 : (x > 100.0 ? 100.0 : x); };
    auto sigmoid = [](double x) { return 1.0 / (1.0 + exp(-x)); };
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
      score_out = va_gt ? (float)sv : 0.0f;
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
