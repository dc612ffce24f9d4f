// BM x 256 tile, four waves, k-slice ring: the encoder GEMM for shapes whose operands are BOTH chunk-major.
//
// Layout.  Activations [M][K] and weights [N][K] are both chunk-major (cm_offset): [rows/32][K/8][32 rows][8 elements].  A 32-row x
// 16-k "piece" (two adjacent chunks) is then 1 KiB CONTIGUOUS in memory: one LDS-DMA instruction (buffer_load_dwordx4 ... lds, lane l
// reads bytes 16 l .. 16 l + 15) copies it unchanged into LDS, where lanes 0-31 / 32-63 of an MFMA fragment read (ds_read_b128) read
// its first / second 512 bytes, row l31 at byte 16 l31: conflict-free without any XOR.  A k-slice (16 k of the whole BM x 256 tile)
// is BM/32 A pieces + 8 B pieces = one ring SLOT; the ring has kSlots of them.
//
// Schedule.  Consumption walks the slices in order; per slice and wave: TM x 4 MFMAs (32x32x16), the fragment reads of the NEXT slice,
// the wave's P pieces of the slice kSlots ahead (LDS-DMA into the slot of the slice everybody has just finished reading), ONE barrier:
//     s_waitcnt vmcnt((kSlots-2) P)  my pieces of slice t+1 have landed (those of slices t+2 .. t+kSlots-1 may still fly)
//             & lgkmcnt(0)           my fragments of slice t are in registers = I am done reading slot t
//     s_barrier                      => slice t+1 is complete in LDS, slot t is free
//     M R M R ... | M M D ...        MFMAs of slice t, reads of slice t+1 first, LDS-DMAs of slice t+kSlots last; the order is pinned
//                                    instruction by instruction (sched_barrier(0)): left alone the scheduler clusters the loads behind
//                                    the MFMAs, where nothing covers their issue time
// The fill stream runs kSlots slices ahead of consumption and straight on across the tile boundaries of the persistent schedule: the
// last kSlots slices of a tile fetch the first kSlots of the block's next tile, and the epilogue runs with (kSlots-1) P pieces in
// flight.  Past the block's last tile the fill stream simply re-reads the first slices of tile (0, 0) into slots nobody consumes (valid
// memory: the scalar offset of a buffer load is not range-checked, so it must not point past the tensor) - the steady state is one
// basic block with exact vmcnt arithmetic, no tail cases.  After an epilogue, slices 0 .. kSlots-2 of the next tile allow kEpiVmem more
// outstanding operations: the epilogue's VMEM operations are younger than the pieces those slices wait for (VMEM operations retire in
// order) and every epilogue issues at least kEpiVmem of them.
//
// Two shapes of the same kernel:
//   BM = 256  one workgroup per CU, one wave per SIMD, wave tile 128 x 128 (256 accumulators, in AGPRs), ring of 8 x 16 KiB.  Reads
//             128 KiB of LDS per K step where the 8-wave ping-pong kernel (bert_gemm.h) reads 192 and synchronises once per 16 MFMAs,
//             but nothing overlaps its epilogue, and one wave per SIMD gets through VALU work (GELU!) at about half the rate of two.
//   BM = 128  TWO workgroups per CU (wave tile 64 x 128, 128 accumulators, ring of 5 x 12 KiB each): the two run unsynchronised, so one
//             workgroup's epilogue overlaps the other's K loop - the matrix pipe always has a K loop to serve.
// On MI355X the matrix pipe is POWER-limited long before it is issue-limited: back-to-back 32x32x16 fp16 MFMAs on random data sustain
// 1.65 PFLOP/s at a clock that has fallen to 1.65 GHz (scripts/ubench/mfma_power.hip), so time saved in the K loop is partly paid back
// as clock; what the epilogue costs, however, is paid in full.
//
// Accumulation order per output element is k ascending in slices of 16, exactly as in the ping-pong kernel: identical bits.
#pragma once
#include "bert_gemm.h"

namespace capamd {

#ifndef CAPAMD_RING_ABLATE
#define CAPAMD_RING_ABLATE 0   // profiling builds only: 1 no LDS-DMA in the loop, 2 no fragment reads
#endif

template <int EPI, typename T, int BM>
struct GemmRing {
  static_assert(BM == 256 || BM == 128, "tile rows");
  using G = GemmKernel<BM, 256, 2, 2, EPI, T>;   // geometry (wave tile BM/2 x 128), tile schedule, LDS-staged epilogues (V^T)
  using CE = CmEpilogue<G, EPI, T>;
  using Lane = typename G::Lane;
  using x8 = typename Half<T>::x8;
  static constexpr int TM = G::TM;                       // 32-row MFMA tiles per wave: 4 / 2
  static constexpr int kPiecesA = BM / 32, kPieces = kPiecesA + 8, P = kPieces / 4;   // 1-KiB pieces of a k-slice; per wave
  static constexpr int kSlotA = kPiecesA * 1024, kSlot = kPieces * 1024;
  static constexpr int kSlots = BM == 256 ? 8 : 5, kRing = kSlot * kSlots;
  static constexpr int kThreads = 256;
  static constexpr int kWgPerCu = BM == 256 ? 1 : 2;
  static constexpr int kLdsBytes = kRing + 4 * G::kEpiLds;
  static constexpr int kEpiVmem = 32;                    // VMEM instructions every epilogue issues per wave, at least
  static constexpr int kVm = (kSlots - 2) * P;           // pieces that may stay in flight at the top of a slice
  static constexpr int kHead = (kSlots - 1) & ~1;        // slices of a tile that may use kVm + kEpiVmem after an epilogue (<= kSlots - 1, even)
  static_assert(kVm + kEpiVmem <= 63 && kLdsBytes * kWgPerCu <= 160 * 1024, "vmcnt immediate / LDS budget");
  static_assert(kPiecesA % 4 == 0, "a wave's p-th piece is an A piece or a B piece for all four waves alike");
  static_assert(EPI != kEpiBiasResidBf16, "the row-major residual epilogue stays on the ping-pong / half-region kernels");

  // s_waitcnt immediate of gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
  static constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

  // scalar byte offsets of this wave's P pieces of k-slice 0 of a tile (a slice further: + 1024).  Piece q = wave + 4 p: the first
  // BM/32 are A row blocks, the rest B row blocks
  struct Bases { unsigned off[P]; };
  static __device__ __forceinline__ Bases bases_of(const GemmArgs& g, int m0, int n0, int wave) {
    Bases b;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int q = wave + 4 * p;
      b.off[p] = 4 * p < kPiecesA ? (unsigned)(((m0 >> 5) + q) * (g.K >> 3)) * 512u : (unsigned)(((n0 >> 5) + q - kPiecesA) * (g.K >> 3)) * 512u;
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
    const int S = g.K >> 4;                          // k-slices per tile (ring_shape: even, >= 16)
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, (int)((size_t)g.M * g.K * 2), 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.W), 0, (int)((size_t)g.N * g.K * 2), 0x00020000);
    const int voff = L.lane * 16;
    // piece p of this wave: LDS-DMA of 1 KiB from byte offset `soff` of its operand into its place in the slot at `dst`
    auto dma = [&](int p, char* dst, unsigned soff) {
      const int q = L.wave + 4 * p;
      if (4 * p < kPiecesA) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(dst + q * 1024), 16, voff, (int)soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(dst + q * 1024), 16, voff, (int)soff, 0, 0);
    };
    // fragment of 32-row block rb of the A (m) / B (n) panel inside a slot: lane (l31, half) reads chunk `half`, row l31
    const int a_base = L.wm * TM * 1024 + L.half * 512 + L.l31 * 16;
    const int b_base = kSlotA + L.wn * 4 * 1024 + L.half * 512 + L.l31 * 16;

    Bases cur = bases_of(g, m0, n0, L.wave);
    // two workgroups per CU: the second half of the grid (the workgroups that land beside the first half) starts about half a tile
    // late, so that the two K loops / epilogues of a CU alternate from the first tile on instead of coinciding
    if (kWgPerCu == 2 && blockIdx.x >= (gridDim.x >> 1))
      for (int i = 0; i < g.ring_stagger; ++i) __builtin_amdgcn_s_sleep(1);
    CAPAMD_STAMP();
#pragma unroll 1
    for (int i = 0; i < kSlots; ++i)
#pragma unroll
      for (int p = 0; p < P; ++p) dma(p, lds + i * kSlot, cur.off[p] + (unsigned)i * 1024u);
    x8 fa[2][TM], fb[2][4];
    auto read_first = [&](int slot) {
#pragma unroll
      for (int j = 0; j < TM; ++j) fa[0][j] = *reinterpret_cast<const x8*>(lds + slot + a_base + j * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i) fb[0][i] = *reinterpret_cast<const x8*>(lds + slot + b_base + i * 1024);
    };
    __builtin_amdgcn_s_waitcnt(waitcnt_imm((kSlots - 1) * P, 15));
    __builtin_amdgcn_s_barrier();
    read_first(0);
    int slot_off = 0;          // byte offset of the slot of the slice being consumed
    bool after_epilogue = false;
    for (int it = 0;; ++it) {
      int m1 = 0, n1 = 0;
      const bool more = G::tile_of(g, it + 1, m1, n1);
      if (!more) { m1 = 0; n1 = 0; }                  // nothing follows: the run-ahead LDS-DMAs re-read the first tile's slices (never consumed)
      const Bases nxt = bases_of(g, m1, n1, L.wave);
      // (no zero fill: the first k-slice of a tile issues its MFMAs with the inline constant 0 as the C operand - every accumulator is
      // written exactly once per slice - instead of 128 / 256 register writes per tile ahead of the K loop; same bits: 0 + a b)
      f32x16 acc[4][TM];
      const bool trans = (EPI == kEpiQkv) && n0 >= 2 * g.H;
      // one k-slice; U = which fragment set holds the slice, VM = outstanding VMEM operations allowed at its top; s = its index in the
      // tile; FIRST: slice 0 of the tile (the accumulators start here)
      auto kslice = [&](auto u_c, auto vm_c, auto tr_c, int s, auto first_c) {
        constexpr int cu = decltype(u_c)::value, nx = cu ^ 1, VM = decltype(vm_c)::value;
        constexpr bool TR = decltype(tr_c)::value, FIRST = decltype(first_c)::value;
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(VM, 0));   // (the builtin, not inline asm: the compiler's own waitcnt pass then knows the
        __builtin_amdgcn_s_barrier();                      // fragment registers are ready and adds no waits of its own in the slice)
        const int nslot = slot_off + kSlot == kRing ? 0 : slot_off + kSlot;
        const char* stn = lds + nslot;
        char* dst = lds + slot_off;
        // the slice kSlots ahead: of this tile, or (the tile's last kSlots slices) of the block's next tile
        const bool own = s + kSlots < S;
        const unsigned so = (unsigned)(own ? s + kSlots : s + kSlots - S) * 1024u;
        unsigned soff[P];
#pragma unroll
        for (int p = 0; p < P; ++p) soff[p] = (own ? cur.off[p] : nxt.off[p]) + so;
        auto M = [&](int idx) {
          const int i = idx / TM, j = idx % TM;
          f32x16 c;
          if constexpr (FIRST) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] = 0.f;
          } else c = acc[i][j];
          acc[i][j] = TR ? Half<T>::mfma(fa[cu][j], fb[cu][i], c) : Half<T>::mfma(fb[cu][i], fa[cu][j], c);
        };
        auto R = [&](int q) {
          if (CAPAMD_RING_ABLATE & 2) return;
          if (q < TM) fa[nx][q] = *reinterpret_cast<const x8*>(stn + a_base + q * 1024);
          else fb[nx][q - TM] = *reinterpret_cast<const x8*>(stn + b_base + (q - TM) * 1024);
        };
        auto D = [&](int p) {
          if (CAPAMD_RING_ABLATE & 1) return;
          dma(p, dst, soff[p]);
        };
        constexpr int NM = 4 * TM, NR = TM + 4;           // 16 / 8 MFMAs, 8 / 6 fragment reads
        constexpr int MPD = (NM - NR) / P;                 // MFMAs per LDS-DMA after the reads: 2 (BM = 256) / 0 (BM = 128: 2 MFMAs, 3 DMAs)
#define CAPAMD_SB __builtin_amdgcn_sched_barrier(0)
        CAPAMD_SB;
#pragma unroll
        for (int q = 0; q < NR; ++q) { M(q); CAPAMD_SB; R(q); CAPAMD_SB; }
        if constexpr (MPD >= 1) {
#pragma unroll
          for (int p = 0; p < P; ++p) {
#pragma unroll
            for (int k = 0; k < MPD; ++k) { M(NR + p * MPD + k); CAPAMD_SB; }
            D(p); CAPAMD_SB;
          }
#pragma unroll
          for (int q = NR + P * MPD; q < NM; ++q) { M(q); CAPAMD_SB; }
        } else {
#pragma unroll
          for (int q = NR; q < NM; ++q) { M(q); CAPAMD_SB; D(q - NR); CAPAMD_SB; }
#pragma unroll
          for (int p = NM - NR; p < P; ++p) { D(p); CAPAMD_SB; }
        }
#undef CAPAMD_SB
        slot_off = nslot;
      };
      using U0 = std::integral_constant<int, 0>;
      using U1 = std::integral_constant<int, 1>;
      using VS = std::integral_constant<int, kVm>;
      using VE = std::integral_constant<int, kVm + kEpiVmem>;
      using F0 = std::false_type;
      using F1 = std::true_type;
      static_assert(kHead >= 2, "the first slice pair of a tile is peeled off the head loop");
      auto tile_loop = [&](auto tr_c) {
        if (after_epilogue) {
          kslice(U0{}, VE{}, tr_c, 0, F1{}); kslice(U1{}, VE{}, tr_c, 1, F0{});
#pragma unroll 1
          for (int s = 2; s < kHead; s += 2) { kslice(U0{}, VE{}, tr_c, s, F0{}); kslice(U1{}, VE{}, tr_c, s + 1, F0{}); }
        } else {
          kslice(U0{}, VS{}, tr_c, 0, F1{}); kslice(U1{}, VS{}, tr_c, 1, F0{});
#pragma unroll 1
          for (int s = 2; s < kHead; s += 2) { kslice(U0{}, VS{}, tr_c, s, F0{}); kslice(U1{}, VS{}, tr_c, s + 1, F0{}); }
        }
#pragma unroll 1
        for (int s = kHead; s < S; s += 2) { kslice(U0{}, VS{}, tr_c, s, F0{}); kslice(U1{}, VS{}, tr_c, s + 1, F0{}); }
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
      // the next tile's first fragments are read again here (the copies the last slice fetched are dead across the epilogue: registers
      // the epilogue needs more than the ~150 cycles this costs per tile)
      read_first(slot_off);
    }
#undef CAPAMD_STAMP
#endif
  }
};

template <int EPI, typename T, int BM>
__global__ __launch_bounds__(256, (BM == 256 ? 1 : 2)) void gemm_ring_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char gemm_ring_lds[];
  GemmRing<EPI, T, BM>::run(a, gemm_ring_lds);
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
