// 256 x 256 tile, four waves, k-slice ring: the encoder GEMM for shapes whose operands are BOTH chunk-major.
//
// Why a second kernel.  The 8-wave ping-pong kernel (bert_gemm.cuh) reads 192 KiB of LDS per K step (a 128 x 64 wave tile re-reads
// every A fragment four times, every B fragment twice) and synchronises eight times per K step; on MI355X the matrix pipe is
// POWER-limited long before it is issue-limited (the shader clock falls from 2.4 GHz to 1.3 - 1.9 GHz as the MFMA duty cycle
// rises: measured in scripts/ubench/gemm4w2.hip with s_memtime against s_memrealtime), so what counts is work per MFMA.  One wave
// per SIMD with a 128 x 128 wave tile (256 accumulator registers, in AGPRs) reads 128 KiB per K step, needs no LDS swizzle, and
// the hardware has no second wave to arbitrate against.
//
// Layout.  Activations [M][K] and weights [N][K] are both chunk-major (cm_offset): [rows/32][K/8][32 rows][8 elements].  A 32-row x
// 16-k "piece" (two adjacent chunks) is then 1 KiB CONTIGUOUS in memory: one LDS-DMA instruction (buffer_load_dwordx4 ... lds, lane l
// reads bytes 16 l .. 16 l + 15) copies it unchanged into LDS, where lanes 0-31 / 32-63 of an MFMA fragment read (ds_read_b128) read
// its first / second 512 bytes, row l31 at byte 16 l31: conflict-free without any XOR.  A k-slice (16 k of the whole 256 x 256 tile)
// is 8 A pieces + 8 B pieces = 16 KiB = one ring SLOT; the ring has 8 slots (two K steps, 128 KiB).
//
// Schedule.  Consumption walks the slices in order; per slice and wave: 16 MFMAs (32x32x16), the 8 fragment reads of the NEXT slice,
// the wave's 4 pieces of the slice 8 ahead (LDS-DMA into the slot of the slice everybody has just finished reading), ONE barrier:
//     s_waitcnt vmcnt(24)   my pieces of slice t+1 have landed (the 6 x 4 pieces of slices t+2 .. t+7 may still fly)
//     s_waitcnt lgkmcnt(0)  my fragments of slice t are in registers = I am done reading slot t
//     s_barrier             => slice t+1 is complete in LDS, slot t is free
//     8 x ds_read_b128 (slice t+1)  |  4 x LDS-DMA (slice t+8 -> slot t)  |  16 x MFMA (slice t)      interleaved by sched_group_barrier
// The fill stream runs 8 slices (two K steps, ~1.5 us) ahead of consumption and straight on across the tile boundaries of the
// persistent schedule: the last 8 slices of a tile fetch the first 8 of the block's next tile, and the epilogue runs with 28 pieces in
// flight.  Past the block's last tile the LDS-DMA source lies beyond the buffer extent (reads as zeros) - the steady state is one basic
// block with exact vmcnt arithmetic, no tail cases.  After an epilogue the first 8 slices allow 32 more outstanding operations
// (the epilogue's stores are younger than the pieces they must not wait for; every epilogue issues at least 32 VMEM instructions).
//
// Accumulation order per output element is k ascending in slices of 16, exactly as in the ping-pong kernel: identical bits.
#pragma once
#include "bert_gemm.cuh"

namespace capamd {

#ifndef CAPAMD_RING_ABLATE
#define CAPAMD_RING_ABLATE 0   // profiling builds only: 1 no LDS-DMA in the loop, 2 no fragment reads, 4 no sched_group_barrier pattern
#endif

template <int EPI, typename T>
struct GemmRing {
  using G = GemmKernel<256, 256, 2, 2, EPI, T>;   // geometry (WMT = WNT = 128, TM = TN = 4), tile schedule, LDS-staged epilogues (V^T)
  using CE = CmEpilogue<G, EPI, T>;
  using Lane = typename G::Lane;
  using x8 = typename Half<T>::x8;
  using x4 = typename Half<T>::x4;
  static constexpr int kSlot = 16 * 1024, kSlots = 8, kRing = kSlot * kSlots;
  static constexpr int kThreads = 256;
  static constexpr int kLdsBytes = kRing + 4 * G::kEpiLds;
  static constexpr int kEpiVmem = 32;   // VMEM instructions every epilogue issues per wave, at least
  static_assert(EPI != kEpiBiasResidBf16, "the row-major residual epilogue stays on the ping-pong / half-region kernels");

  struct Bases { unsigned a[2], w[2]; };   // scalar byte offsets of this wave's pieces of k-slice 0 of a tile (a slice further: + 1024)

  static __device__ __forceinline__ Bases bases_of(const GemmArgs& g, int m0, int n0, int wave) {
    Bases b;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      b.a[p] = (unsigned)(((m0 >> 5) + wave * 2 + p) * (g.K >> 3)) * 512u;
      b.w[p] = (unsigned)(((n0 >> 5) + wave * 2 + p) * (g.K >> 3)) * 512u;
    }
    return b;
  }

  static __device__ __forceinline__ void run(const GemmArgs& g, char* lds) {
#if defined(__HIP_DEVICE_COMPILE__)
    Lane L;
    L.tid = threadIdx.x; L.lane = L.tid & 63; L.wave = __builtin_amdgcn_readfirstlane(L.tid >> 6);
    L.wm = L.wave >> 1; L.wn = L.wave & 1; L.l31 = L.lane & 31; L.half = L.lane >> 5;
    unsigned long long* dbg = g.dbg ? g.dbg + (size_t)blockIdx.x * 32 : nullptr;
    int dbg_i = 0;
#define CAPAMD_STAMP() do { if (dbg && L.tid == 0 && dbg_i < 32) dbg[dbg_i++] = __builtin_readcyclecounter(); } while (0)
    int m0, n0;
    if (!G::tile_of(g, 0, m0, n0)) return;
    const int S = g.K >> 4;                          // k-slices per tile (>= 16: ring_shape)
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, (int)((size_t)g.M * g.K * 2), 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.W), 0, (int)((size_t)g.N * g.K * 2), 0x00020000);
    const int voff = L.lane * 16;
    auto issue = [&](const Bases& b, int s, int slot_off) {
      char* d = lds + slot_off;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int rb = L.wave * 2 + p;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(d + rb * 1024), 16, voff, (int)(b.a[p] + (unsigned)s * 1024u), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(d + 8192 + rb * 1024), 16, voff, (int)(b.w[p] + (unsigned)s * 1024u), 0, 0);
      }
    };
    // fragment of 32-row block rb of the A (m) / B (n) panel inside a slot: lane (l31, half) reads chunk `half`, row l31
    const int a_base = L.wm * 4 * 1024 + L.half * 512 + L.l31 * 16;
    const int b_base = 8192 + L.wn * 4 * 1024 + L.half * 512 + L.l31 * 16;

    Bases cur = bases_of(g, m0, n0, L.wave);
    CAPAMD_STAMP();
#pragma unroll 1
    for (int i = 0; i < kSlots; ++i) issue(cur, i, i * kSlot);
    x8 fa[2][4], fb[2][4];
    wait_vmcnt<28>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[0][i] = *reinterpret_cast<const x8*>(lds + a_base + i * 1024);
      fb[0][i] = *reinterpret_cast<const x8*>(lds + b_base + i * 1024);
    }
    int slot_off = 0;          // byte offset of the slot of the slice being consumed
    bool after_epilogue = false;
    for (int it = 0;; ++it) {
      int m1 = 0, n1 = 0;
      const bool more = G::tile_of(g, it + 1, m1, n1);
      if (!more) { m1 = g.M; n1 = g.N; }              // beyond both extents: the run-ahead LDS-DMAs read zeros
      const Bases nxt = bases_of(g, m1, n1, L.wave);
      f32x16 acc[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      const bool trans = (EPI == kEpiQkv) && n0 >= 2 * g.H;
      // one k-slice (see the header); U = which fragment set holds the slice, VM = outstanding VMEM operations allowed at its top
      auto kslice = [&](auto u_c, auto vm_c, auto tr_c, const Bases& src, int fs) {
        constexpr int cu = decltype(u_c)::value, nx = cu ^ 1, VM = decltype(vm_c)::value;
        constexpr bool TR = decltype(tr_c)::value;
        wait_vmcnt<VM>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char* stn = lds + ((slot_off + kSlot) & (kRing - 1));
        if (!(CAPAMD_RING_ABLATE & 2)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            fa[nx][i] = *reinterpret_cast<const x8*>(stn + a_base + i * 1024);
            fb[nx][i] = *reinterpret_cast<const x8*>(stn + b_base + i * 1024);
          }
        }
        if (!(CAPAMD_RING_ABLATE & 1)) issue(src, fs, slot_off);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = TR ? Half<T>::mfma(fa[cu][j], fb[cu][i], acc[i][j]) : Half<T>::mfma(fb[cu][i], fa[cu][j], acc[i][j]);
#if !(CAPAMD_RING_ABLATE & 4)
        // 16 MFMA | 8 DS read | 4 VMEM  ->  (M R M V M R M) x 4
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
#endif
        slot_off = (slot_off + kSlot) & (kRing - 1);
      };
      using U0 = std::integral_constant<int, 0>;
      using U1 = std::integral_constant<int, 1>;
      using V24 = std::integral_constant<int, 24>;
      using VE = std::integral_constant<int, 24 + kEpiVmem>;
      auto tile_loop = [&](auto tr_c) {
        // head: slices 0 .. 7 (refilled with 8 .. 15 of this tile); after an epilogue its VMEM operations are younger than the pieces
        if (after_epilogue) {
#pragma unroll 1
          for (int s = 0; s < 8; s += 2) { kslice(U0{}, VE{}, tr_c, cur, s + 8); kslice(U1{}, VE{}, tr_c, cur, s + 9); }
        } else {
#pragma unroll 1
          for (int s = 0; s < 8; s += 2) { kslice(U0{}, V24{}, tr_c, cur, s + 8); kslice(U1{}, V24{}, tr_c, cur, s + 9); }
        }
#pragma unroll 1
        for (int s = 8; s < S - 8; s += 2) { kslice(U0{}, V24{}, tr_c, cur, s + 8); kslice(U1{}, V24{}, tr_c, cur, s + 9); }
        // tail: the last 8 slices fetch slices 0 .. 7 of the block's next tile
#pragma unroll 1
        for (int s = 0; s < 8; s += 2) { kslice(U0{}, V24{}, tr_c, nxt, s); kslice(U1{}, V24{}, tr_c, nxt, s + 1); }
      };
      if (trans) tile_loop(std::true_type{});
      else tile_loop(std::false_type{});
      CAPAMD_STAMP();
      char* wl = lds + kRing + L.wave * G::kEpiLds;
      typename Half<T>::x4 rs[1][1][4];
      if constexpr (EPI == kEpiResidStats) CE::epilogue_cm_resid(g, m0, n0, L, acc);
      else if (trans) G::template epilogue<true>(g, wl, m0, n0, L, acc, rs);
      else if (g.out_cm) CE::epilogue_cm(g, m0, n0, L, acc);
      else G::template epilogue<false>(g, wl, m0, n0, L, acc, rs);
      CAPAMD_STAMP();
      if (!more) break;
      after_epilogue = true;
      cur = nxt;
      m0 = m1; n0 = n1;
    }
#undef CAPAMD_STAMP
#endif
  }
};

template <int EPI, typename T>
__global__ __launch_bounds__(256, 1) void gemm_ring_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char gemm_ring_lds[];
  GemmRing<EPI, T>::run(a, gemm_ring_lds);
}

// [rows][K] row-major -> chunk-major (weights, once per model): one thread per 16-byte chunk
template <typename T>
__global__ void to_chunk_major_kernel(const T* __restrict__ src, T* __restrict__ dst, int rows, int K) {
  const int64_t nchunk = (int64_t)rows * (K >> 3);
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk; c += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = c / (K >> 3);
    const int kc = (int)(c - r * (K >> 3));
    *reinterpret_cast<uint4*>(dst + cm_offset(r, kc * 8, K)) = *reinterpret_cast<const uint4*>(src + r * K + kc * 8);
  }
}

}  // namespace capamd
