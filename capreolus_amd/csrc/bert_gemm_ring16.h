// 256 x 256 tile, four waves, the k-slice ring of bert_gemm_ring.h consumed by v_mfma_f32_16x16x32 (round 5).
//
// Why another K loop.  Every 32x32x16 kernel of this library lands at the same 0.95-1.1 PFLOP/s on the encoder's shapes whatever its
// schedule, at clocks between 1.06 and 1.5 GHz (profiles/r05/ring_timeline.txt): the part is power-limited and cycles saved come back as
// clock.  The vendor's kernel is 6-12 % faster on the same shapes with 16x16x32 MFMAs; scripts/ubench/gemm4w5.hip - this file's K loop
// on its own, next to gemm4w2.hip, the 32x32x16 ring - measured why that is not the epilogue: the SAME ring, the same LDS bytes per
// flop, one barrier per 1024 MFMA cycles in both, and the K step (64 k of a 256 x 256 tile) takes 1.78 us instead of 1.97 (QKV), 1.30
// instead of 1.42 (output projection), 1.73 instead of 1.91 (FFN1), 2.03 instead of 2.06 (FFN2) at the same clock
// (profiles/r05/gemm4w5_vs_4w2.txt).  Per flop a 16x16x32 MFMA moves half the accumulator registers of a 32x32x16 one (4 in, 4 out per
// 16 k-flops against 16 and 16 per 32 k) for twice the operand registers: 0.25 against 0.31 register bytes per flop.
//
// Layout.  Operands chunk-major as in bert_gemm_ring.h; a ring SLOT is a k-slice of 16 (8 A pieces + 8 B pieces of 1 KiB), a STEP is
// 32 k = two adjacent slots (the ring's 8 slots = 4 steps).  An operand fragment of a 16-row tile is 16 rows x 32 k: lane (l15 = lane
// % 16, q = lane / 16) reads row l15 of k-chunk q - chunks 0, 1 in the step's first slot, 2, 3 in its second - one ds_read_b128, the 16
// lanes of a q 256 contiguous bytes: conflict-free.  A wave owns 128 x 128 = 8 x 8 tiles of 16 x 16, 64 accumulators of 4 registers
// (AGPRs; the MFMAs are inline assembly accumulating in place - left to the register allocator the 64 four-register tuples get copied
// around); with the weight fragment as the A operand lane (l15, q) holds, of tile (i, j), row m = 16 j + l15, columns n = 16 i + 4 q
// + 0..3.  Accumulation order per output element: k ascending in steps of 32, the 32 products of a step summed by the MFMA - NOT the
// bits of the 32x32x16 kernels (k in steps of 16); every launch of a given GEMM of the encoder takes the same kernel whatever its M.
//
// Schedule.  Per step and wave: 64 MFMAs, the 16 fragment reads of the next step, the wave's 8 pieces of the step four ahead (into the
// two slots every wave has just finished reading), ONE barrier; issue order pinned: M M R M M, a DMA after every second group.  vmcnt
// arithmetic as in the 32x32x16 ring with a step as the unit: 16 pieces (two steps) may stay in flight at the top of a step, 32 more
// in the two steps after an epilogue.  The first step of a tile writes the accumulators (C = 0): no zero fill.
//
// Epilogues.  The chunk-major ones of bert_gemm.h re-derived for the 16 x 16 layout: a lane owns 4 consecutive columns of a row; the
// two lanes that hold the halves of an 8-column chunk are 16 apart, and the two 16-row tiles of a 32-row block sit in the same lanes -
// v_permlane16_swap on the PACKED registers of tile (i, 2 jp) and (i, 2 jp + 1) leaves lane q even with the whole chunk of row l15 and
// lane q odd with the whole chunk of row 16 + l15: one 16-byte store per lane, 1 KiB contiguous per instruction.  V^T tiles (QKV): the
// operand roles swapped (lane <-> n, registers <-> 4 consecutive keys), regrouped through the wave's 4 KiB of LDS into 128-byte row
// segments as before.
#pragma once
#include "bert_gemm_ring.h"

namespace capamd {

typedef __attribute__((ext_vector_type(4))) float f32x4;

#ifndef CAPAMD_R16_ABL
#define CAPAMD_R16_ABL 0   // profiling builds of the residual epilogue only: 1 no residual loads, 2 no statistics, 4 no stores, 8 no exchange
#endif

template <typename T>
struct Mfma16;
template <>
struct Mfma16<__bf16> {
  using x8 = typename Half<__bf16>::x8;
  static __device__ __forceinline__ void first(f32x4& c, x8 a, x8 b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b)); }
  static __device__ __forceinline__ void acc(f32x4& c, x8 a, x8 b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
};
template <>
struct Mfma16<_Float16> {
  using x8 = typename Half<_Float16>::x8;
  static __device__ __forceinline__ void first(f32x4& c, x8 a, x8 b) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b)); }
  static __device__ __forceinline__ void acc(f32x4& c, x8 a, x8 b) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
};

// BM = 256: one workgroup per CU (wave tile 128 x 128, ring of 4 steps).  BM = 128: TWO workgroups per CU (wave tile 64 x 128: 32
// accumulators of 4 AGPRs + 96 fragment registers of a wave's 256; ring of 3 steps x 24 KiB) that run unsynchronised, so one workgroup's
// epilogue - FFN1's erf-GELU is 15.8 k cycles on one wave per SIMD beside a 30.5 k K loop - overlaps the other's K loop; no LDS staging
// bytes, so no V^T tiles (QKV stays on 256 rows).
template <int EPI, typename T, int BM = 256>
struct GemmRing16 {
  static_assert(BM == 256 || BM == 128, "tile rows");
  using G = GemmKernel<BM, 256, 2, 2, EPI, T>;    // tile schedule (tile_of), V^T addressing (out_offset), staging bytes
  using x8 = typename Half<T>::x8;
  using x4 = typename Half<T>::x4;
  static constexpr int WM = BM / 2;                          // rows of a wave's tile
  static constexpr int NI = 8, NJ = WM / 16;                 // 16 x 16 tiles of a wave: i along n, j along m
  static constexpr int kPiecesA = BM / 32, kPieces = kPiecesA + 8, P = kPieces / 4;   // 1-KiB pieces of a k-slice; per wave
  static constexpr int kSlotA = kPiecesA * 1024, kSlot = kPieces * 1024, kSteps = BM == 256 ? 4 : 3, kSlots = 2 * kSteps, kRing = kSlot * kSlots;
  static constexpr int kThreads = 256;
  static constexpr int kWgPerCu = BM == 256 ? 1 : 2;
  static constexpr int kLdsBytes = kRing + (BM == 256 ? 4 * G::kEpiLds : 0);
  static constexpr int kEpiVmem = 32;                        // VMEM instructions every epilogue issues per wave, at least
  static constexpr int kVm = (kSteps - 2) * 2 * P;           // pieces of the youngest kSteps - 2 steps may stay in flight at the top of a step
  static_assert(kPiecesA % 4 == 0, "a wave's p-th piece is an A piece or a B piece for all four waves alike");
  static_assert(BM == 256 || EPI != kEpiQkv, "V^T tiles are regrouped through LDS staging bytes only the 256-row tile has");
  static_assert(kVm + kEpiVmem <= 63 && kLdsBytes * kWgPerCu <= 160 * 1024, "vmcnt immediate / LDS budget");
  static_assert(EPI == kEpiBiasBf16 || EPI == kEpiBiasGeluBf16 || EPI == kEpiQkv || EPI == kEpiResidStats, "epilogues of the fused encoder");

  static constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

  struct Lane {
    int tid, lane, wave, wm, wn, l15, q;
  };
  struct Bases { unsigned off[P]; };
  static __device__ __forceinline__ Bases bases_of(const GemmArgs& g, int m0, int n0, int wave) {
    Bases b;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int pc = wave + 4 * p;     // piece: the first kPiecesA are A row blocks, the other 8 B row blocks
      b.off[p] = 4 * p < kPiecesA ? (unsigned)(((m0 >> 5) + pc) * (g.K >> 3)) * 512u : (unsigned)(((n0 >> 5) + pc - kPiecesA) * (g.K >> 3)) * 512u;
    }
    return b;
  }

  // lanes of row 1 / 3 (lanes 16..31, 48..63) of x <-> lanes of row 0 / 2 of y   (wait states by hand, as swap32 in bert_gemm.h)
  static __device__ __forceinline__ void swap16(unsigned& x, unsigned& y) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  }
  static __device__ __forceinline__ void pin_group(f32x4 (&row)[NJ]) {   // keeps a column group's tiles in AGPRs up to this point
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("" : "+a"(row[j]));
  }

  // ---- epilogue: bias / folded LayerNorm (+ Q / 8, + GELU) into a chunk-major output ------------------------------------------------
  static __device__ __forceinline__ void epilogue_cm(const GemmArgs& a, int m0, int n0, const Lane& L, f32x4 (&acc)[NI][NJ]) {
    T* base = static_cast<T*>(a.out_bf16);
    int ncols = a.N, nloc = n0 + L.wn * 128;
    float scale = 1.f;
    if (EPI == kEpiQkv) {
      ncols = a.H;
      if (n0 >= a.H) { base = static_cast<T*>(a.out_k); nloc -= a.H; }
      else scale = 0.125f;
    }
    const float es = EPI == kEpiBiasGeluBf16 ? 0.5f : scale;     // folded into the affine form (bert_gemm.h: epilogue_cm)
    const bool ln = a.ln_mu != nullptr;
    float2 mr[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      mr[j] = ln ? a.ln_mr[m0 + L.wm * WM + j * 16 + L.l15] : make_float2(0.f, 1.f);
      mr[j].y *= es;
    }
    const float* csp = ln ? a.ln_cs : a.bias;      // (mu = 0 without folded LayerNorm: any finite vector serves as cs)
    float4 b4s[2], cs4s[2];
    auto load_cols = [&](int i, int buf) {
      const int n = n0 + L.wn * 128 + i * 16 + 4 * L.q;
      b4s[buf] = *reinterpret_cast<const float4*>(a.bias + n);
      cs4s[buf] = *reinterpret_cast<const float4*>(csp + n);
    };
    load_cols(0, 0);
    const int nch = ncols >> 3;
    // byte offset of this lane's 16-byte piece of (column group 0, row block 0) from `base`; a column group further + 1024, a 32-row
    // block further + jstride: 32-bit arithmetic on a wave-uniform base (a tensor stays below 4 GiB: ring_shape)
    const unsigned lane_off = (unsigned)((((m0 + L.wm * WM) >> 5) * nch + (nloc >> 3) + (L.q >> 1)) * 32 + (L.q & 1) * 16 + L.l15) * 16u;
    const unsigned jstride = (unsigned)nch * 512u;
    char* const baseb = reinterpret_cast<char*>(base);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < NI) load_cols(i + 1, (i + 1) & 1);      // one group ahead, before this group's stores (in-order VMEM retirement)
      pin_group(acc[i]);
      __builtin_amdgcn_sched_barrier(0);
      float4 b4 = b4s[i & 1];
      b4.x *= es; b4.y *= es; b4.z *= es; b4.w *= es;
      const float4 cs4 = cs4s[i & 1];
#pragma unroll
      for (int jp = 0; jp < NJ / 2; ++jp) {
        unsigned pk[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = 2 * jp + h;
          const float mu = mr[j].x, rs = mr[j].y;
          f32x2 v01 = {rs * (acc[i][j][0] - mu * cs4.x) + b4.x, rs * (acc[i][j][1] - mu * cs4.y) + b4.y};
          f32x2 v23 = {rs * (acc[i][j][2] - mu * cs4.z) + b4.z, rs * (acc[i][j][3] - mu * cs4.w) + b4.w};
          if (EPI == kEpiBiasGeluBf16) { v01 = gelu_erf_half2(v01); v23 = gelu_erf_half2(v23); }
          const x4 o = {(T)v01.x, (T)v01.y, (T)v23.x, (T)v23.y};
          const uint2 u = __builtin_bit_cast(uint2, o);
          pk[h][0] = u.x; pk[h][1] = u.y;
        }
        // even q: row l15 of the 32-row block, columns 8 (q / 2) .. + 7 (its own 4 + the odd lane's); odd q: row 16 + l15, same chunk
        swap16(pk[0][0], pk[1][0]);
        swap16(pk[0][1], pk[1][1]);
        *reinterpret_cast<uint4*>(baseb + (size_t)(lane_off + (unsigned)i * 1024u + (unsigned)jp * jstride)) = make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
      }
    }
  }

  // ---- epilogue: pre-LayerNorm sum with the residual re-normalised on the fly + row statistics (bert_gemm.h: epilogue_cm_resid) ---
  static __device__ __forceinline__ void epilogue_cm_resid(const GemmArgs& a, int m0, int n0, const Lane& L, f32x4 (&acc)[NI][NJ]) {
    T* base = static_cast<T*>(a.out_bf16);
    const T* rsrc = static_cast<const T*>(a.res_src);
    const int nch = a.N >> 3, nloc = n0 + L.wn * 128, nslot = a.N >> 6;
    float rrs[NJ], rc[NJ];
    f32x2 s1[2][NJ / 2], s2[2][NJ / 2];       // per 64-column slot and 32-row block: (even, odd) element partial sums of this lane's row
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float2 mr = a.res_mr[m0 + L.wm * WM + j * 16 + L.l15];
      rrs[j] = mr.y;
      rc[j] = -mr.x * mr.y;
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int jp = 0; jp < NJ / 2; ++jp) s1[sl][jp] = s2[sl][jp] = f32x2{0.f, 0.f};
    // The residual comes from HBM / the Infinity Cache (a 98 MB tensor another kernel wrote) and nothing else runs on this SIMD: a piece
    // requested one step (16 values of arithmetic) ahead arrives 20-30 steps' worth of cycles later, and the epilogue - 31 k cycles
    // against the K loop's 30 k at K = 768 (profiles/r05/ring16_timeline.txt) - was a chain of 32 memory latencies.  kResAhead steps
    // are in flight: the first kResAhead requested together at the top, piece t + kResAhead when step t's registers are free.
    constexpr int kResSteps = NI * NJ / 2, kResAhead = kResSteps < 16 ? kResSteps : 16;
    float4 bbs[2], ggs[2];
    x4 r4s[kResAhead][2];
    auto load_cols = [&](int i, int buf) {
      const int n = n0 + L.wn * 128 + i * 16 + 4 * L.q;
      bbs[buf] = *reinterpret_cast<const float4*>(a.bias + n);
      ggs[buf] = *reinterpret_cast<const float4*>(a.res_gamma + n);
    };
    // byte offsets from the wave-uniform tensor bases (32-bit: a tensor stays below 4 GiB); a column group further + 1024, a 32-row
    // block further + jstride
    const unsigned blk_off = (unsigned)(((m0 + L.wm * WM) >> 5) * nch + (nloc >> 3) + (L.q >> 1)) * 512u;
    const unsigned res_off = blk_off + (unsigned)L.l15 * 16u + 8u * (unsigned)(L.q & 1);             // the lane's own 8 bytes of row l15
    const unsigned out_off = blk_off + (unsigned)((L.q & 1) * 16 + L.l15) * 16u;                     // its 16-byte piece after the exchange
    const unsigned jstride = (unsigned)nch * 512u;
    const char* const rsrcb = reinterpret_cast<const char*>(rsrc);
    char* const baseb = reinterpret_cast<char*>(base);
    auto load_res = [&](int t, int buf) {   // t = i * (NJ / 2) + jp: the residual where the accumulators put the value - row l15 (tile 2 jp) and 16 + l15
      const int i = t / (NJ / 2), jp = t % (NJ / 2);
      const char* rp = rsrcb + (size_t)(res_off + (unsigned)i * 1024u + (unsigned)jp * jstride);
      if (CAPAMD_R16_ABL & 1) { r4s[buf][0] = r4s[buf][1] = x4{(T)1.f, (T)1.f, (T)1.f, (T)1.f}; return; }
      r4s[buf][0] = *reinterpret_cast<const x4*>(rp);
      r4s[buf][1] = *reinterpret_cast<const x4*>(rp + 256);
    };
    load_cols(0, 0);
#pragma unroll
    for (int t = 0; t < kResAhead; ++t) load_res(t, t);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < NI) load_cols(i + 1, (i + 1) & 1);
      pin_group(acc[i]);
      __builtin_amdgcn_sched_barrier(0);
      const float4 bb = bbs[i & 1], gg = ggs[i & 1];
      const f32x2 b01 = {bb.x, bb.y}, b23 = {bb.z, bb.w}, g01 = {gg.x, gg.y}, g23 = {gg.z, gg.w};
#pragma unroll
      for (int jp = 0; jp < NJ / 2; ++jp) {
        const int t = i * (NJ / 2) + jp;
        unsigned pk[2][2];
        uint2 rus[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) rus[h] = __builtin_bit_cast(uint2, r4s[t % kResAhead][h]);
        if (t + kResAhead < kResSteps) load_res(t + kResAhead, t % kResAhead);     // (its registers were just read)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = 2 * jp + h;
          const uint2 ru = rus[h];
          const f32x2 rr = {rrs[j], rrs[j]}, cc = {rc[j], rc[j]};
          const f32x2 v01 = f32x2{acc[i][j][0], acc[i][j][1]} + b01, v23 = f32x2{acc[i][j][2], acc[i][j][3]} + b23;
          const f32x2 t01 = Half<T>::unpack2(ru.x) * rr + cc, t23 = Half<T>::unpack2(ru.y) * rr + cc;
          const f32x2 o01 = t01 * g01 + v01, o23 = t23 * g23 + v23;             // fp32 sum, ONE rounding
          const x4 o = {(T)o01.x, (T)o01.y, (T)o23.x, (T)o23.y};
          const uint2 u = __builtin_bit_cast(uint2, o);
          pk[h][0] = u.x; pk[h][1] = u.y;
        }
        if (!(CAPAMD_R16_ABL & 8)) {
          swap16(pk[0][0], pk[1][0]);
          swap16(pk[0][1], pk[1][1]);
        }
        if (!(CAPAMD_R16_ABL & 4) || L.lane == 99)
        *reinterpret_cast<uint4*>(baseb + (size_t)(out_off + (unsigned)i * 1024u + (unsigned)jp * jstride)) = make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
        // statistics of what the consumers will read, from the registers as stored: 8 values of ONE row (l15, or 16 + l15 for odd q)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (CAPAMD_R16_ABL & 2) { s1[i >> 2][jp][0] += __uint_as_float(pk[k >> 1][k & 1]); continue; }
          const f32x2 qv = Half<T>::unpack2(pk[k >> 1][k & 1]);
          s1[i >> 2][jp] += qv;
          s2[i >> 2][jp] = qv * qv + s2[i >> 2][jp];
        }
      }
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int jp = 0; jp < NJ / 2; ++jp) {
        // lanes q and q + 2 hold the two chunks of the same row: add them; lanes 0..31 are the 32 rows of the block
        const float p1 = s1[sl][jp].x + s1[sl][jp].y, p2 = s2[sl][jp].x + s2[sl][jp].y;
        const float t1 = p1 + __shfl_xor(p1, 32, 64), t2 = p2 + __shfl_xor(p2, 32, 64);
        const int mrow = m0 + L.wm * WM + jp * 32 + (L.lane & 31);
        if (L.lane < 32) *reinterpret_cast<float2*>(a.stat_part + ((int64_t)mrow * nslot + ((n0 + L.wn * 128) >> 6) + sl) * 2) = make_float2(t1, t2);
      }
  }

  // ---- epilogue: V^T tiles of the QKV projection (operand roles swapped: lane <-> n = l15, registers <-> keys m = 4 q + 0..3) --------
  static __device__ __forceinline__ void epilogue_vt(const GemmArgs& a, char* wl, int m0, int n0, const Lane& L, f32x4 (&acc)[NI][NJ]) {
    const bool ln = a.ln_mu != nullptr;
    float bias_t[NI], cs_t[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      bias_t[i] = a.bias[n0 + L.wn * 128 + i * 16 + L.l15];
      cs_t[i] = ln ? a.ln_cs[n0 + L.wn * 128 + i * 16 + L.l15] : 0.f;
    }
    T* base = static_cast<T*>(a.out_vt);
    typename G::Lane GL;
    GL.wm = L.wm; GL.wn = L.wn;
    // a round: 32 rows of n (tiles 2 ip, 2 ip + 1) x 64 keys (tiles 4 jq .. 4 jq + 3) = 32 x 128 bytes in the wave's LDS
#pragma unroll
    for (int ip = 0; ip < NI / 2; ++ip) {
      __builtin_amdgcn_sched_barrier(0);
      pin_group(acc[2 * ip]);
      pin_group(acc[2 * ip + 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jq = 0; jq < NJ / 4; ++jq) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int j = 4 * jq + jj;
          float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = make_float4(1.f, 1.f, 1.f, 1.f);
          if (ln) {
            const int mrow = m0 + L.wm * WM + j * 16 + 4 * L.q;
            mu = *reinterpret_cast<const float4*>(a.ln_mu + mrow);
            rs = *reinterpret_cast<const float4*>(a.ln_rstd + mrow);
          }
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * ip + ii;
            const float b = bias_t[i], cs = cs_t[i];
            const x4 o = {(T)(rs.x * (acc[i][j][0] - mu.x * cs) + b), (T)(rs.y * (acc[i][j][1] - mu.y * cs) + b),
                          (T)(rs.z * (acc[i][j][2] - mu.z * cs) + b), (T)(rs.w * (acc[i][j][3] - mu.w * cs) + b)};
            const int row = ii * 16 + L.l15, c16 = jj * 2 + (L.q >> 1);
            *reinterpret_cast<x4*>(wl + row * 128 + ((c16 ^ (row & 7)) << 4) + (L.q & 1) * 8) = o;
          }
        }
        // (no wait between a wave's own LDS writes and reads: its DS operations execute in issue order)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = L.lane + 64 * k, row = c >> 3, ch = c & 7;
          const uint4 x = *reinterpret_cast<const uint4*>(wl + row * 128 + ((ch ^ (row & 7)) << 4));
          *reinterpret_cast<uint4*>(base + G::template out_offset<true>(a, m0, n0, GL, ip, row, jq * 64 + ch * 8)) = x;
        }
        asm volatile("" ::: "memory");  // the next round's writes stay behind these reads
      }
    }
  }

  static __device__ __forceinline__ void run(const GemmArgs& g, char* lds) {
#if defined(__HIP_DEVICE_COMPILE__)
    Lane L;
    L.tid = threadIdx.x; L.lane = L.tid & 63; L.wave = __builtin_amdgcn_readfirstlane(L.tid >> 6);
    L.wm = L.wave >> 1; L.wn = L.wave & 1; L.l15 = L.lane & 15; L.q = L.lane >> 4;
#ifdef CAPAMD_PROFILING      // (cycle stamps per tile: the profiling build only - the pointer and its index cost the 128-row form two spills)
    unsigned long long* dbg = g.dbg ? g.dbg + (size_t)blockIdx.x * 32 : nullptr;
    int dbg_i = 0;
#define CAPAMD_STAMP() do { if (dbg && L.tid == 0 && dbg_i < 32) dbg[dbg_i++] = __builtin_readcyclecounter(); } while (0)
#else
#define CAPAMD_STAMP() do { } while (0)
#endif
    int m0, n0;
    if (!G::tile_of(g, 0, m0, n0)) return;
    const int S2 = g.K >> 5;                         // steps per tile (launch_gemm: even, >= 8)
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, (int)((size_t)g.M * g.K * 2), 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.W), 0, (int)((size_t)g.N * g.K * 2), 0x00020000);
    const int voff = L.lane * 16;
    auto dma = [&](int p, char* slot, unsigned soff) {
      const int pc = L.wave + 4 * p;
      if (4 * p < kPiecesA) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(slot + pc * 1024), 16, voff, (int)soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(slot + pc * 1024), 16, voff, (int)soff, 0, 0);
    };
    // fragment of 16-row tile t of the A (m) / B (n) panel inside a step: + (t >> 1) * 1024 + (t & 1) * 256
    const int a_base = L.wm * (WM / 32) * 1024 + (L.q & 1) * 512 + L.l15 * 16 + (L.q >> 1) * kSlot;
    const int b_base = kSlotA + L.wn * 4 * 1024 + (L.q & 1) * 512 + L.l15 * 16 + (L.q >> 1) * kSlot;

    // kEpiResidStats: the epilogue reads the tile's 128 KiB of residual, and every CU reaches its epilogue at about the same time - 32 MB
    // requested at once from a tensor another kernel wrote ~1 GB of traffic ago: HBM-bound, 11 k of the epilogue's 20 k cycles
    // (profiles/r05/resid16_ablation.txt).  GemmArgs::res_touch (CAPAMD_R16_TOUCH=1; NOT the default) lets the first four steps of a tile
    // TOUCH the wave's part of it - 4 x 8 KiB contiguous, one 128-byte line per lane, as a 4-byte LDS-DMA into the (otherwise unused)
    // staging bytes: no destination register to keep alive - so that the lines come in under the K loop.  Measured: the epilogue 20 k ->
    // 15 k cycles, the K loop 31.5 k -> 38.5 k (VMEM operations retire in order: a miss to HBM in the queue holds up the counted waits
    // for the L2-hit pieces behind it) - a loss (profiles/r05/ring16_timeline_touch.txt).
    const auto rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(EPI == kEpiResidStats ? g.res_src : g.A), 0,
                                                        (int)((size_t)g.M * (EPI == kEpiResidStats ? g.N : g.K) * 2), 0x00020000);
    auto touch_residual = [&](int blk, int tm0, int tn0) {
      const unsigned soff = (unsigned)((((tm0 + L.wm * WM + blk * 32) >> 5) * (g.N >> 3) + ((tn0 + L.wn * 128) >> 3)) * 32) * 16u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rres, (lds_void_t*)(lds + kRing + L.wave * G::kEpiLds + blk * 256), 4, L.lane * 128, (int)soff, 0, 0);
    };
    Bases cur = bases_of(g, m0, n0, L.wave);
    CAPAMD_STAMP();
#pragma unroll 1
    for (int i = 0; i < kSlots; ++i)
#pragma unroll
      for (int p = 0; p < P; ++p) dma(p, lds + i * kSlot, cur.off[p] + (unsigned)i * 1024u);
    x8 fa[2][NJ], fb[2][NI];
    auto read_first = [&](int slot) {
#pragma unroll
      for (int t = 0; t < NJ; ++t) fa[0][t] = *reinterpret_cast<const x8*>(lds + slot + a_base + (t >> 1) * 1024 + (t & 1) * 256);
#pragma unroll
      for (int t = 0; t < NI; ++t) fb[0][t] = *reinterpret_cast<const x8*>(lds + slot + b_base + (t >> 1) * 1024 + (t & 1) * 256);
    };
    __builtin_amdgcn_s_waitcnt(waitcnt_imm((kSlots - 2) * P, 15));     // the first step's two slices have landed
    __builtin_amdgcn_s_barrier();
    read_first(0);
    int slot_off = 0;          // byte offset of the first slot of the step being consumed
    bool after_epilogue = false;
    for (int it = 0;; ++it) {
      int m1 = 0, n1 = 0;
      const bool more = G::tile_of(g, it + 1, m1, n1);
      if (!more) { m1 = 0; n1 = 0; }                  // nothing follows: the run-ahead LDS-DMAs re-read the first tile's slices (never consumed)
      const Bases nxt = bases_of(g, m1, n1, L.wave);
      f32x4 acc[NI][NJ];
      const bool trans = (EPI == kEpiQkv) && n0 >= 2 * g.H;
      // one step (32 k); U: which fragment set holds it, VM: outstanding VMEM operations allowed at its top, s: its index in the tile
      auto kstep = [&](auto u_c, auto vm_c, auto tr_c, int s, auto first_c) {
        constexpr int cu = decltype(u_c)::value, nx = cu ^ 1, VM = decltype(vm_c)::value;
        constexpr bool TR = decltype(tr_c)::value, FIRST = decltype(first_c)::value;
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(VM, 0));
        __builtin_amdgcn_s_barrier();
        const int nslot = slot_off + 2 * kSlot == kRing ? 0 : slot_off + 2 * kSlot;
        const char* stn = lds + nslot;
        char* dst = lds + slot_off;
        // the step four ahead: of this tile, or (the tile's last four steps) of the block's next tile
        const bool own = s + kSteps < S2;
        const unsigned so = (unsigned)(own ? s + kSteps : s + kSteps - S2) * 2048u;
        unsigned soff[P];
#pragma unroll
        for (int p = 0; p < P; ++p) soff[p] = (own ? cur.off[p] : nxt.off[p]) + so;
#define CAPAMD_M16(idx)                                                                  \
  do {                                                                                   \
    constexpr int i_ = (idx) / NJ, j_ = (idx) % NJ;                                      \
    if (FIRST) { if (TR) Mfma16<T>::first(acc[i_][j_], fa[cu][j_], fb[cu][i_]); else Mfma16<T>::first(acc[i_][j_], fb[cu][i_], fa[cu][j_]); } \
    else { if (TR) Mfma16<T>::acc(acc[i_][j_], fa[cu][j_], fb[cu][i_]); else Mfma16<T>::acc(acc[i_][j_], fb[cu][i_], fa[cu][j_]); }            \
  } while (0)
#define CAPAMD_SB __builtin_amdgcn_sched_barrier(0)
        auto R = [&](int r) {
          if (r < NJ) fa[nx][r] = *reinterpret_cast<const x8*>(stn + a_base + (r >> 1) * 1024 + (r & 1) * 256);
          else fb[nx][r - NJ] = *reinterpret_cast<const x8*>(stn + b_base + ((r - NJ) >> 1) * 1024 + ((r - NJ) & 1) * 256);
        };
        auto D = [&](int d) {       // piece d of the step's 2 P: slice d / P, the wave's piece d % P
          dma(d % P, dst + (d / P) * kSlot, soff[d % P] + (unsigned)(d / P) * 1024u);
        };
        CAPAMD_SB;
#define CAPAMD_G16(g_)                                                                                                    \
  CAPAMD_M16(4 * (g_)); CAPAMD_SB; CAPAMD_M16(4 * (g_) + 1); CAPAMD_SB; R(g_); CAPAMD_SB; CAPAMD_M16(4 * (g_) + 2); CAPAMD_SB; \
  CAPAMD_M16(4 * (g_) + 3); CAPAMD_SB;                                                                                   \
  if ((g_) & 1) { D((g_) >> 1); CAPAMD_SB; }
        // 128 rows: 32 MFMAs, 12 reads, 6 DMAs in eight groups - M M R M M, a second read in the first four groups, a DMA from the third on
#define CAPAMD_G16H(g_)                                                                                                   \
  CAPAMD_M16(4 * (g_)); CAPAMD_SB; CAPAMD_M16(4 * (g_) + 1); CAPAMD_SB; R((g_) < 4 ? 2 * (g_) : 4 + (g_)); CAPAMD_SB;    \
  CAPAMD_M16(4 * (g_) + 2); CAPAMD_SB; CAPAMD_M16(4 * (g_) + 3); CAPAMD_SB;                                              \
  if ((g_) < 4) { R(2 * (g_) + 1); CAPAMD_SB; }                                                                           \
  if ((g_) >= 2) { D((g_) - 2); CAPAMD_SB; }
        if constexpr (BM == 256) {
          CAPAMD_G16(0) CAPAMD_G16(1) CAPAMD_G16(2) CAPAMD_G16(3) CAPAMD_G16(4) CAPAMD_G16(5) CAPAMD_G16(6) CAPAMD_G16(7)
          CAPAMD_G16(8) CAPAMD_G16(9) CAPAMD_G16(10) CAPAMD_G16(11) CAPAMD_G16(12) CAPAMD_G16(13) CAPAMD_G16(14) CAPAMD_G16(15)
        } else {
          CAPAMD_G16H(0) CAPAMD_G16H(1) CAPAMD_G16H(2) CAPAMD_G16H(3) CAPAMD_G16H(4) CAPAMD_G16H(5) CAPAMD_G16H(6) CAPAMD_G16H(7)
        }
        // (one more VMEM operation in flight in steps 0..3: the counted waits above only get stricter by it - everything older than the
        // youngest kVm operations has landed, and those are a subset of what is younger than the awaited step)
        if (BM == 256 && EPI == kEpiResidStats && g.res_touch && s < 4) { touch_residual(s, m0, n0); CAPAMD_SB; }
#undef CAPAMD_G16H
#undef CAPAMD_G16
#undef CAPAMD_SB
#undef CAPAMD_M16
        slot_off = nslot;
      };
      using U0 = std::integral_constant<int, 0>;
      using U1 = std::integral_constant<int, 1>;
      using VS = std::integral_constant<int, kVm>;
      using VE = std::integral_constant<int, kVm + kEpiVmem>;
      using F0 = std::false_type;
      using F1 = std::true_type;
      auto tile_loop = [&](auto tr_c) {
        // (after an epilogue the first two steps wait for pieces older than the epilogue's >= kEpiVmem VMEM operations: steps 1 and 2
        // of the tile were issued before it)
        if (after_epilogue) { kstep(U0{}, VE{}, tr_c, 0, F1{}); kstep(U1{}, VE{}, tr_c, 1, F0{}); }
        else { kstep(U0{}, VS{}, tr_c, 0, F1{}); kstep(U1{}, VS{}, tr_c, 1, F0{}); }
#pragma unroll 1
        for (int s = 2; s < S2; s += 2) { kstep(U0{}, VS{}, tr_c, s, F0{}); kstep(U1{}, VS{}, tr_c, s + 1, F0{}); }
      };
      if (EPI == kEpiQkv && trans) tile_loop(std::true_type{});
      else tile_loop(std::false_type{});
      // the last MFMAs' results are read by the epilogue's v_accvgpr_read: the wait states the assembler would insert behind a
      // matrix instruction it can see (inline assembly hides them from its hazard recogniser)
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
      CAPAMD_STAMP();
      if constexpr (EPI == kEpiResidStats) epilogue_cm_resid(g, m0, n0, L, acc);
      else if (EPI == kEpiQkv && trans) epilogue_vt(g, lds + kRing + L.wave * G::kEpiLds, m0, n0, L, acc);
      else epilogue_cm(g, m0, n0, L, acc);
      CAPAMD_STAMP();
      if (!more) break;
      after_epilogue = true;
      cur = nxt;
      m0 = m1; n0 = n1;
      read_first(slot_off);
    }
#undef CAPAMD_STAMP
#endif
  }
};

template <int EPI, typename T, int BM>
__global__ __launch_bounds__(256, (BM == 256 ? 1 : 2)) void gemm_ring16_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char gemm_ring16_lds[];
  GemmRing16<EPI, T, BM>::run(a, gemm_ring16_lds);
}

}  // namespace capamd
