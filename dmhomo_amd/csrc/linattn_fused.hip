// K3f — LinearAttention with its PreNorm LayerNorm and to_qkv projection fused in (CFG:96-103, 246-269):
// q, k, v are never written to HBM.
//
//   pass 1  linattn_kv_kernel     x -> LN -> k, v = W_kv LN(x) (registers) -> per split: max_n k, sum exp, exp(k)^T v
//   (merge  linattn_merge_kernel  of attention.hip: finishes the softmax over the n pixels, scales by 1/n)
//   pass 2  linattn_qo_kernel     x -> LN -> q = W_q LN(x) -> softmax over d, * scale -> out = q ctx   [B][n][128]
//
// against  LayerNorm (r+w) + to_qkv conv (w 384 channels) + context (r 256) + apply (r 128, w 128):  per 128x128
// attention of the bench (50 rows) 0.84 GB of HBM traffic instead of 3.6 GB.
//
// The projections are f16x3 GEMMs exactly as in conv_f16x3.hip (block-scaled fp16 pieces of the fp32 operands,
// three v_mfma_f32_16x16x32_f16 per product block, fp32 accumulation; see that file for the error analysis); the
// staged operand is the LayerNorm output (x - mean) * rstd * g, computed as chan_layernorm_kernel computes it from
// the per-pixel (mean, rstd) of dmh_pixel_stats, under a STATIC block scale (sqrt(C) * max|g| bounds it).  One wave per
// head; 64 pixels per sub-tile.  The two attention products run on the fp16 matrix cores as well (round 2):
//   pass 1 multiplies pixels x channels ("pixels as rows"): the accumulator layout (lane = channel, registers =
//          pixels) is at once the A operand (exp(k - m), split into two fp16 pieces) and the B operand (v, per-column
//          scale, two pieces) of the 16x16x32 MFMAs that contract over the pixels — no lane movement between the GEMMs;
//   pass 2 multiplies channels x pixels (q^T): its accumulators (lane = pixel, registers = d), softmaxed and split, are
//          the B operand of out^T = ctx^T q^T, whose A operand (the head's context) is split once per workgroup.
#include <stdlib.h>

#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define LA_PART (32 + 32 + 1024)  // per (b, split, head): max[32], sum[32], ctx[32][32]  (as attention.hip)

// Diagnostic builds only (make stamps; tools/la_ablate.py -> profiles/r05_linattn_ablation.txt): pieces of the two passes
// switched off by a device word, so that the time each one is worth is MEASURED before anything is rebuilt around it.
//   bit 0  the second fp16 piece of every split is not formed (cvt only: the split's fma_mix / cvt pairs are gone)
//   bit 1  no exponentials (the softmax arguments pass through unchanged)
//   bit 2  pass 2: no LayerNorm + residual + store behind the head exchange
//   bit 3  every load of x / statistics comes from the sample's first pixels (one cached line: no HBM stream)
//   bit 4  no projection MFMAs (staging, softmax, products and stores stay)
// The product build compiles every LA_ABL(...) to 0.
#ifdef DMH_STAMPS
__device__ int g_la_ablate = 0;
extern "C" int dmh_la_set_ablate(int v) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_la_ablate), &v, sizeof(int)) == hipSuccess ? 0 : -1;
}
#define LA_ABL_LOAD() const int la_abl = g_la_ablate
#define LA_ABL(bit) ((la_abl >> (bit)) & 1)
#else
#define LA_ABL_LOAD() constexpr int la_abl = 0
#define LA_ABL(bit) 0
#endif

namespace {
constexpr int KC = 32;      // channels per chunk = K of one MFMA

// the split of common.h, or (diagnostic bit 0) its first half only
__device__ __forceinline__ void la_split2(float x0, float x1, float s, unsigned& h, unsigned& r, int cheap) {
  if (cheap) {
    float t0 = x0 * s, t1 = x1 * s;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(t0), "v"(t1));
    r = 0u;
  } else {
    dmh_split2(x0, x1, s, h, r);
  }
}
__device__ __forceinline__ void la_split8(const float (&x)[8], float s, dmh_half8& h, dmh_half8& r, int cheap) {
  if (cheap) {
    uint4 hh, rr = make_uint4(0u, 0u, 0u, 0u);
    la_split2(x[0], x[1], s, hh.x, rr.x, 1);
    la_split2(x[2], x[3], s, hh.y, rr.y, 1);
    la_split2(x[4], x[5], s, hh.z, rr.z, 1);
    la_split2(x[6], x[7], s, hh.w, rr.w, 1);
    asm volatile("s_nop 3" : "+v"(hh.x), "+v"(hh.y), "+v"(hh.z), "+v"(hh.w));
    h = __builtin_bit_cast(dmh_half8, hh);
    r = __builtin_bit_cast(dmh_half8, rr);
  } else {
    dmh_split8(x, s, h, r);
  }
}
constexpr int PITCH = 160;  // LDS bytes per staged pixel (conv_f16x3.hip: conflict-free ds_read_b128)
constexpr int TP = 64;      // pixels per sub-tile
constexpr int TILE_BYTES = TP * PITCH;

__device__ __forceinline__ unsigned absbits(float x) { return __float_as_uint(x) & 0x7fffffffu; }

// Stages one 32-channel chunk of a 64-pixel sub-tile: global x -> LayerNorm -> block scale -> two fp16 planes in LDS.
// All 256 threads; thread = (pixel tid>>3 (+32), channel quad tid&7).
//
// The block scale is STATIC (round 2): |(x - mean) * rstd| <= sqrt(C - 1) for any C numbers, so sqrt(C) * max|g| bounds
// every staged value, and with the second piece stored unscaled (conv_f16x3.hip) an over-estimated scale costs nothing
// until it is 2^18 too large — so there is no block maximum to find: no reduction, no LDS atomic, no barrier in front of the
// split, no accumulator rescale between chunks.  NBUF == 2 alternates two LDS tiles, which leaves ONE barrier per chunk
// (tile written -> tile read; a wave writes tile q + 2 only after the barrier of q + 1, which every wave passes after its
// reads of q); NBUF == 1 (the fully fused pass 2, whose LDS is taken by the head-exchange buffer) keeps the second one.
template <int NBUF>
struct Stager {
  const float* xb;     // x of this sample
  const float* g;      // LayerNorm gain
  int C, n, p0;
  int c4, pix0;
  float mean[2], rstd[2];
  bool ok[2];
  float4 v[2];
  float4 gq;
  float sc, inv_sc;
  int seq = 0;  // staged chunks so far (selects the LDS tile; keeps alternating across sub-tiles)
  int abl = 0;  // diagnostic builds: the LA_ABL bits (0 in the product)

  // every wave derives the same scale from g: bound = sqrt(C) * max|g|, bound * sc in [2^14, 2^15)
  __device__ __forceinline__ void init_scale() {
    float m = 0.f;
    for (int i = threadIdx.x & 63; i < C; i += 64) m = fmaxf(m, fabsf(g[i]));
#pragma unroll
    for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    const float bound = m * sqrtf((float)C);
    const int e = min(max((int)(__builtin_amdgcn_readfirstlane(__float_as_uint(bound)) >> 23) & 0xff, 16), 254);
    sc = __uint_as_float((unsigned)(268 - e) << 23);
    inv_sc = __uint_as_float((unsigned)(e - 14) << 23);
  }
  __device__ __forceinline__ void begin_tile(const float* stats_b, int p0_) {
    p0 = p0_;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pix = p0 + pix0 + 32 * i;
      ok[i] = pix < n;
      const int pc = (abl & 8) ? pix0 : (ok[i] ? pix : n - 1);
      mean[i] = stats_b[(size_t)pc * 2 + 0];
      rstd[i] = stats_b[(size_t)pc * 2 + 1];
    }
  }
  __device__ __forceinline__ void issue(int ch) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pix = p0 + pix0 + 32 * i;
      const int pc = (abl & 8) ? pix0 : (pix < n ? pix : n - 1);  // clamped address, masked value
      v[i] = ld4(xb + (size_t)pc * C + ch * KC + c4 * 4);
    }
    gq = ld4(g + ch * KC + c4 * 4);
  }
  // returns the LDS tile the chunk was staged into
  __device__ __forceinline__ unsigned char* stage(unsigned char* tiles) {
    unsigned char* tile = tiles + (NBUF == 2 ? (seq & 1) * TILE_BYTES : 0);
    ++seq;
    if (NBUF == 1) __syncthreads();  // every wave is done reading the previous chunk's tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 x = v[i];
      if (ok[i]) {
        x.x = (x.x - mean[i]) * rstd[i] * gq.x;  // exactly chan_layernorm_kernel's expression
        x.y = (x.y - mean[i]) * rstd[i] * gq.y;
        x.z = (x.z - mean[i]) * rstd[i] * gq.z;
        x.w = (x.w - mean[i]) * rstd[i] * gq.w;
      } else {
        x = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      uint2 h1, h2;                                        // second piece stored unscaled (conv_f16x3.hip)
      la_split2(x.x, x.y, sc, h1.x, h2.x, abl & 1);
      la_split2(x.z, x.w, sc, h1.y, h2.y, abl & 1);
      unsigned char* dst = tile + (pix0 + 32 * i) * PITCH + c4 * 8;
      *reinterpret_cast<uint2*>(dst) = h1;
      *reinterpret_cast<uint2*>(dst + 64) = h2;
    }
    __syncthreads();
    return tile;
  }
  __device__ __forceinline__ float inv_scale() const { return inv_sc; }
};

// ---- RingStager (round 3, pass 1 at C == 64): raw x reaches the workgroup by LDS-DMA, a whole sub-tile ahead, and a
// sub-tile is staged in ONE pass.
// Round 2 staged from registers one 32-channel chunk ahead and one chunk at a time: per sub-tile two latency chains (load ->
// LayerNorm -> split -> LDS write; ~750 cycles each for ~40 instructions: tools/kv_stamps.py) and two barriers, 8 KB in flight
// per workgroup.  Here a unit = one sub-tile ([64 pixels][64 channels] fp32 = 16 KB, contiguous in NHWC) is fetched by
// global_load_lds_dwordx4 — 16 pieces of 1 KB, four per wave, no VGPRs, counted on vmcnt — into a ring of two units, with the
// sub-tile's 64 (mean, rstd) pairs (global_load_lds_dword, 16 pixels per wave) in front of it.  The staging pass reads its
// four 16 B runs per thread from the raw unit, applies LayerNorm and the fp16 split exactly as Stager does, and writes BOTH
// chunks' staged tiles — into one of two tile SETS, so that one raw s_barrier per sub-tile is enough: it says the staged set
// is written, every wave's pieces of the NEXT unit have landed (each wave waits vmcnt(0) in front of it: the next unit is all
// it has outstanding; a __syncthreads() is avoided only because its fence would also be emitted where nothing needs it),
// and the unit just read is free for the DMA of unit t + 2.  No compiler-visible vector-memory load is left in the loop
// (hipcc would wait vmcnt(0) for it wherever its result is used): the LayerNorm gains sit in registers, the statistics come
// through the ring, the kernel consumes every other load result in front of the loop.
// What it bought (profiles/r03_linattn_notes.txt): deep prefetch alone (six chunk-units in flight, the first form) 162 -> 155
// us — the pass was not waiting for HBM; see the notes for what it is waiting for.
// s_barrier without __syncthreads()'s fences (which would also wait vmcnt(0) wherever they are emitted).  The intrinsic is
// declared as touching no memory, so hipcc may move LDS accesses across it: the empty asm statements pin them (round 3:
// without the second one the fragment reads behind the barrier were free to move above it wherever no asm statement — a DMA
// issue — followed: tests/test_gpu_soak.py caught the kernel computing from half-written tiles, 58 of 60 launches).
__device__ __forceinline__ void dmh_raw_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

struct RingStager {
  static constexpr int RD = 2;
  static constexpr int UNIT = TP * 64 * 4;            // 16384: one sub-tile of 64 channels
  static constexpr int STATS_SLOTS = 4;               // sub-tiles whose statistics may be in the ring at once (3) rounded up
  static constexpr int BYTES = RD * UNIT + STATS_SLOTS * TP * 8;
  const float* xb;
  const float* stats_b;
  int n, p0w, U;               // pixels per sample, first pixel of the workgroup, sub-tiles of the workgroup
  int c4, pix0, lane, wave;
  unsigned char* ring;
  unsigned ring_lds;
  float4 gq[2];
  float sc, inv_sc;
  int seq = 0;
  int abl = 0;  // diagnostic builds: the LA_ABL bits (0 in the product)

  __device__ __forceinline__ void init(const float* x_b, const float* stats_b_, const float* g, int n_, int p0w_, int tiles,
                                       unsigned char* ring_) {
    xb = x_b;
    stats_b = stats_b_;
    n = n_;
    p0w = p0w_;
    const int left = (n - p0w + TP - 1) / TP;
    U = left < tiles ? left : tiles;
    lane = threadIdx.x & 63;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c4 = threadIdx.x & 7;
    pix0 = threadIdx.x >> 3;
    ring = ring_;
    ring_lds = (unsigned)(size_t)ring_;
    float m = fabsf(g[lane]);                         // C == 64: one gain per lane
#pragma unroll
    for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    const float bound = m * 8.0f;                     // sqrt(64) * max |g|  (Stager::init_scale)
    const int e = min(max((int)(__builtin_amdgcn_readfirstlane(__float_as_uint(bound)) >> 23) & 0xff, 16), 254);
    sc = __uint_as_float((unsigned)(268 - e) << 23);
    inv_sc = __uint_as_float((unsigned)(e - 14) << 23);
    gq[0] = ld4(g + c4 * 4);
    gq[1] = ld4(g + KC + c4 * 4);
  }
  __device__ __forceinline__ void glds(const void* src, unsigned lds_dst, bool x4) {
    unsigned keep;
    if (x4)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
    else
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
  }
  // sub-tile t (< U): its statistics, then the wave's four pieces (16 pixels x 256 B)
  __device__ __forceinline__ void issue_unit(int t) {
    const int p0 = (abl & 8) ? 0 : p0w + t * TP;
    const int fl = min((p0 + wave * 16) * 2 + (lane & 31), 2 * n - 1);        // clamped; pixels beyond n are masked later
    if (lane < 32) glds(stats_b + fl, ring_lds + RD * UNIT + (t % STATS_SLOTS) * (TP * 8) + wave * 128, false);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int piece = wave * 4 + j;                 // pixels 4 * piece .. + 3 of the sub-tile, 256 B each
      const int pc = min(p0 + piece * 4 + (lane >> 4), n - 1);
      glds(xb + (size_t)pc * 64 + (lane & 15) * 4, ring_lds + (t % RD) * UNIT + piece * 1024, true);
    }
  }
  __device__ __forceinline__ void prime() {
    if (U > 0) issue_unit(0);
    if (U > 1) {
      issue_unit(1);
      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // unit 1's five pieces may stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    dmh_raw_barrier();
  }
  // stage sub-tile `seq` into tile set seq & 1: chunk 0 at the returned address, chunk 1 TILE_BYTES behind it
  __device__ __forceinline__ unsigned char* stage(unsigned char* tiles) {
    const int t = seq++;
    unsigned char* set = tiles + (t & 1) * (2 * TILE_BYTES);
    const unsigned char* raw = ring + (t % RD) * UNIT;
    const int p0 = p0w + t * TP;
    const float* st = reinterpret_cast<const float*>(ring + RD * UNIT + (t % STATS_SLOTS) * (TP * 8));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float2 mr = *reinterpret_cast<const float2*>(st + (pix0 + 32 * i) * 2);
      const bool ok = p0 + pix0 + 32 * i < n;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        float4 x = *reinterpret_cast<const float4*>(raw + (pix0 + 32 * i) * 256 + ch * 128 + c4 * 16);
        const float4 gv = gq[ch];
        if (ok) {
          x.x = (x.x - mr.x) * mr.y * gv.x;  // exactly chan_layernorm_kernel's expression
          x.y = (x.y - mr.x) * mr.y * gv.y;
          x.z = (x.z - mr.x) * mr.y * gv.z;
          x.w = (x.w - mr.x) * mr.y * gv.w;
        } else {
          x = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        uint2 h1, h2;
        la_split2(x.x, x.y, sc, h1.x, h2.x, abl & 1);
        la_split2(x.z, x.w, sc, h1.y, h2.y, abl & 1);
        unsigned char* dst = set + ch * TILE_BYTES + (pix0 + 32 * i) * PITCH + c4 * 8;
        *reinterpret_cast<uint2*>(dst) = h1;
        *reinterpret_cast<uint2*>(dst + 64) = h2;
      }
    }
    return set;
  }
  // after stage() of sub-tile t: the staged set is written (and the reads of raw unit t are back: the writes depend on them);
  // wait for this wave's pieces of unit t + 1 (all it has in flight), ONE barrier, refill the slot every wave has finished reading
  __device__ __forceinline__ void sync() {
    const int t = seq - 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dmh_raw_barrier();
    if (t + RD < U) issue_unit(t + RD);
  }
  __device__ __forceinline__ float inv_scale() const { return inv_sc; }
};

}  // namespace

// ------------------------------------------------------------------------------------------ pass 1
// grid = B * nsplit workgroups; workgroup (b, sp) covers `tiles` sub-tiles of 64 pixels; wave = head.
// wkv: [head 4][chunk][nb 4: k lo, k hi, v lo, v hi][plane 2][lane 64] x 16 B, then 256 floats 2^-k per (head, nb, col)
// RES: number of channel chunks whose weight fragments stay in registers for the whole workgroup (2 when C == 64: the
// 128x128 level, where a workgroup walks 4 sub-tiles and round 1 re-fetched the head's 16 KB of fragments from L2 for each
// of them); 0: the fragments of a chunk are fetched while the chunk is staged.
template <int RES>
__global__ __launch_bounds__(256, 2) void linattn_kv_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                            const float* __restrict__ g, const uint4* __restrict__ wkv,
                                                            const float* __restrict__ oscale, float* __restrict__ partial,
                                                            int n, int C, int nsplit, int tiles,
                                                            const int32_t* __restrict__ rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* tiles_lds = smem;  // two staging tiles

  const int jb = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  if (rows && jb >= rows[0]) return;   // (a row subset, common.h: the workgroups of inactive rows retire)
  const int b = dmh_rows_phys(rows, jb);
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  const int nch = C / KC;

  LA_ABL_LOAD();
  Stager<2> st;
  st.abl = la_abl;
  st.xb = x + (size_t)b * n * C;
  st.g = g;
  st.C = C;
  st.n = n;
  st.c4 = tid & 7;
  st.pix0 = tid >> 3;
  st.init_scale();
  const float* stats_b = stats + (size_t)b * n * 2;

  const uint4* wb = wkv + (size_t)h * nch * (8 * 64) + lane;
  float osc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) osc[nb] = oscale[(h * 4 + nb) * 16 + l15];

  // running state over the sub-tiles: per column d = db*16 + l15 (lanes; the maxima in the base-2 domain), and the context
  // TRANSPOSED, ctx[eb][db] = block (rows e = eb*16 + 4*kg + r, columns d = db*16 + l15): the softmax rescale of a column d
  // is then one multiplier per lane (round 2, late; with rows d it took eight cross-lane reads per sub-tile)
  float m_run[2] = {-INFINITY, -INFINITY}, s_run[2] = {0.f, 0.f};
  float4v ctx[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) ctx[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

  if (sp * tiles * TP < n) {
    st.begin_tile(stats_b, sp * tiles * TP);
    st.issue(0);
  }
  uint4 wres[RES ? RES : 1][8];
  if (RES) {
#pragma unroll
    for (int ch = 0; ch < RES; ++ch)
#pragma unroll
      for (int i = 0; i < 8; ++i) wres[ch][i] = wb[(size_t)(ch * 8 + i) * 64];
  }
  for (int tI = 0; tI < tiles; ++tI) {
    const int p0 = (sp * tiles + tI) * TP;
    if (p0 >= n) break;
    float4v acc[4][4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = float4v{0.f, 0.f, 0.f, 0.f};

    auto chunk = [&](int ch, const uint4 (&bq)[8]) __attribute__((always_inline)) {
      const unsigned char* tile = st.stage(tiles_lds);
      if (ch + 1 < nch) st.issue(ch + 1);
      half8 a[4][2];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          a[mb][pl] = *reinterpret_cast<const half8*>(tile + (mb * 16 + l15) * PITCH + kg * 16 + pl * 64);
#define LA_TERM(pl, bexpr)                                                                          \
  _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) _Pragma("unroll") for (int nb = 0; nb < 4; ++nb) \
      acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mb][pl], bexpr, acc[mb][nb], 0, 0, 0);
      if (!LA_ABL(4)) {
      LA_TERM(1, __builtin_bit_cast(half8, bq[nb * 2]))
      LA_TERM(0, __builtin_bit_cast(half8, bq[nb * 2 + 1]))
      LA_TERM(0, __builtin_bit_cast(half8, bq[nb * 2]))
      } else {
        acc[0][0][0] += (float)a[0][0][0] + (float)a[3][1][7];   // (keeps the fragment reads alive)
      }
#undef LA_TERM
    };
    if (RES) {
#pragma unroll
      for (int ch = 0; ch < RES; ++ch) chunk(ch, wres[ch]);
    } else {
      for (int ch = 0; ch < nch; ++ch) {
        uint4 bq[8];  // this chunk's weight fragments travel while the chunk is staged
#pragma unroll
        for (int i = 0; i < 8; ++i) bq[i] = wb[(size_t)(ch * 8 + i) * 64];
        chunk(ch, bq);
      }
    }
    const float inv_s = st.inv_scale();
    if (tI + 1 < tiles && p0 + TP < n) {  // the next sub-tile's first chunk travels during the softmax / context math
      st.begin_tile(stats_b, p0 + TP);
      st.issue(0);
    }

    // ---- k, v of this sub-tile: acc[mb][nb][r] = value(pixel p0 + mb*16 + 4*kg + r, column nb*16 + l15)
    // (1 / block scale and the weight column's 2^-k are both powers of two: one exact multiplier)
    const bool full = p0 + TP <= n;  // uniform: only the last sub-tile of a ragged image masks its pixels
    float m_new[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const float kk = inv_s * osc[db] * 1.44269504088896341f;   // k * log2(e): the softmax runs in the base-2 domain
      float m = -INFINITY;
      if (full) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float kv = acc[mb][db][r] * kk;
            acc[mb][db][r] = kv;
            m = fmaxf(m, kv);
          }
      } else {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool valid = p0 + mb * 16 + 4 * kg + r < n;
            const float kv = valid ? acc[mb][db][r] * kk : -INFINITY;
            acc[mb][db][r] = kv;
            m = fmaxf(m, kv);
          }
      }
      m = rows_max(m);
      m_new[db] = fmaxf(m_run[db], m);
    }
    // rescale the running context columns by 2^(m_run - m_new): column d = this lane
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const float fcol = __builtin_amdgcn_exp2f(m_run[db] - m_new[db]);  // 0 on the first sub-tile (m_run = -inf)
      s_run[db] *= fcol;
      ctx[0][db] *= fcol;
      ctx[1][db] *= fcol;
      m_run[db] = m_new[db];
    }
    // ---- ctx[d][e] += sum_n p[n][d] v[n][e] on the fp16 matrix cores (round 2; round 1: 64 fp32 16x16x4 MFMAs per sub-tile,
    // which execute on the vector ALUs).  The accumulator layout is the operand layout: lane (l15, kg) holds column l15
    // (d for p, e for v) of the pixels mb*16 + 4*kg + r, and K slot 8*kg + j of step s is pixel (2s + (j >> 2))*16 + 4*kg
    // + (j & 3) for A and B alike.  p = 2^(k - m) lies in [0, 1]: * 2^10, split; v: one power-of-two scale per wave and
    // sub-tile, split.  v is the A operand: D rows = e, D columns = d = l15 (the transposed context, see above).
    half8 p1[2][2], p2[2][2], v1[2][2], v2[2][2];  // [block][K step]
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float arg = acc[2 * ks + (j >> 2)][db][j & 3] - m_new[db];
          pv[j] = LA_ABL(1) ? arg : __builtin_amdgcn_exp2f(arg);  // 2^-inf = 0 for pixels beyond n
          s += pv[j];
        }
        la_split8(pv, 1024.f, p1[db][ks], p2[db][ks], LA_ABL(0));
      }
      s = rows_sum(s);
      s_run[db] += s;
    }
    // v: one power-of-two scale for the wave's 64 pixels x 32 columns (fp16 pieces are floating point: a block-wide
    // scale costs range, not precision — as in the conv kernels), so the product is unscaled by one uniform factor
    unsigned mx = 0u;
#pragma unroll
    for (int eb = 0; eb < 2; ++eb) {
      // (pixels beyond n were staged as zeros, so their v is exactly 0 without a mask)
      const float kv2 = inv_s * osc[2 + eb];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float vv = acc[mb][2 + eb][r] * kv2;
          acc[mb][2 + eb][r] = vv;
          mx = max(mx, absbits(vv));
        }
    }
    mx = wave_max_u32(mx);
    const int exv = min(max((int)(mx >> 23), 32), 254);
    const float scv = __uint_as_float((unsigned)(268 - exv) << 23);         // block maximum * scv in [2^14, 2^15)
    const float inv_v = __uint_as_float((unsigned)(exv - 14 - 10) << 23);   // 1 / scv, and the 2^10 of p
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float vv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) vv[j] = acc[2 * ks + (j >> 2)][2 + eb][j & 3];
        la_split8(vv, scv, v1[eb][ks], v2[eb][ks], LA_ABL(0));
      }
    float4v t[2][2];  // [eb][db]: four independent chains, term by term; A = v (rows e), B = p (columns d)
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int db = 0; db < 2; ++db)
        t[eb][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v2[eb][0], p1[db][0], float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int db = 0; db < 2; ++db) t[eb][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v2[eb][1], p1[db][1], t[eb][db], 0, 0, 0);
#pragma unroll
    for (int st2 = 0; st2 < 2; ++st2)
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
#pragma unroll
        for (int db = 0; db < 2; ++db)
          t[eb][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1[eb][st2], p2[db][st2], t[eb][db], 0, 0, 0);
#pragma unroll
    for (int st2 = 0; st2 < 2; ++st2)
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
#pragma unroll
        for (int db = 0; db < 2; ++db)
          t[eb][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1[eb][st2], p1[db][st2], t[eb][db], 0, 0, 0);
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int db = 0; db < 2; ++db) ctx[eb][db] += t[eb][db] * inv_v;
  }

  float* out = partial + ((size_t)(b * nsplit + sp) * 4 + h) * LA_PART;
  if (kg == 0) {
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      out[db * 16 + l15] = m_run[db] * 0.693147180559945309f;   // back to the natural domain the merge works in
      out[32 + db * 16 + l15] = s_run[db];
    }
  }
  // ctx[d][e] row-major: this lane's four consecutive e of column d are one 16 B piece
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
      st4(out + 64 + (db * 16 + l15) * 32 + eb * 16 + 4 * kg,
          make_float4(ctx[eb][db][0], ctx[eb][db][1], ctx[eb][db][2], ctx[eb][db][3]));
}

__global__ __launch_bounds__(256, 2) void linattn_kv_ring_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                            const float* __restrict__ g, const uint4* __restrict__ wkv,
                                                            const float* __restrict__ oscale, float* __restrict__ partial,
                                                            int n, int C, int nsplit, int tiles,
                                                            const int32_t* __restrict__ rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* tiles_lds = smem;  // two sets of two staging tiles, then the raw ring
  constexpr int RES = 2;            // C == 64: both chunks' weight fragments stay in registers

  const int jb = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  if (rows && jb >= rows[0]) return;   // (a row subset, common.h: the workgroups of inactive rows retire)
  const int b = dmh_rows_phys(rows, jb);
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  const int l15 = lane & 15, kg = lane >> 4;

  LA_ABL_LOAD();
  RingStager st;
  st.abl = la_abl;
  st.init(x + (size_t)b * n * C, stats + (size_t)b * n * 2, g, n, sp * tiles * TP, tiles, smem + 4 * TILE_BYTES);
  const int nch = 2;

  const uint4* wb = wkv + (size_t)h * nch * (8 * 64) + lane;
  float osc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) osc[nb] = oscale[(h * 4 + nb) * 16 + l15];

  // running state over the sub-tiles: per column d = db*16 + l15 (lanes; the maxima in the base-2 domain), and the context
  // TRANSPOSED, ctx[eb][db] = block (rows e = eb*16 + 4*kg + r, columns d = db*16 + l15): the softmax rescale of a column d
  // is then one multiplier per lane (round 2, late; with rows d it took eight cross-lane reads per sub-tile)
  float m_run[2] = {-INFINITY, -INFINITY}, s_run[2] = {0.f, 0.f};
  float4v ctx[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) ctx[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

  uint4 wres[RES ? RES : 1][8];
  if (RES) {
#pragma unroll
    for (int ch = 0; ch < RES; ++ch)
#pragma unroll
      for (int i = 0; i < 8; ++i) wres[ch][i] = wb[(size_t)(ch * 8 + i) * 64];
  }
  // every compiler-visible vector-memory load of the kernel is consumed HERE, in front of the ring: hipcc cannot see the
  // LDS-DMA, and a load result first used inside the loop would get an s_waitcnt vmcnt(0) there that drains the ring on
  // every unit (it did: the LayerNorm gains)
#pragma unroll
  for (int ch = 0; ch < RES; ++ch)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(wres[ch][i].x), "v"(wres[ch][i].y), "v"(wres[ch][i].z), "v"(wres[ch][i].w));
  asm volatile("" ::"v"(st.gq[0].x), "v"(st.gq[0].y), "v"(st.gq[0].z), "v"(st.gq[0].w), "v"(st.gq[1].x), "v"(st.gq[1].y),
               "v"(st.gq[1].z), "v"(st.gq[1].w), "v"(osc[0]), "v"(osc[1]), "v"(osc[2]), "v"(osc[3]));
  st.prime();
#ifdef DMH_STAMPS
  // diagnostic build only (make stamps; tools/kv_stamps.py): per-wave cycle totals of each phase, written over the partial
  unsigned long long tk[6] = {0, 0, 0, 0, 0, 0}, t_prev, t_now;
#define KSTAMP(i)                                                                \
  __builtin_amdgcn_sched_barrier(0);                                             \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_now)::"memory"); \
  __builtin_amdgcn_sched_barrier(0);                                             \
  tk[i] += t_now - t_prev;                                                       \
  t_prev = t_now;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
#else
#define KSTAMP(i)
#endif
  for (int tI = 0; tI < tiles; ++tI) {
    const int p0 = (sp * tiles + tI) * TP;
    if (p0 >= n) break;
    float4v acc[4][4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = float4v{0.f, 0.f, 0.f, 0.f};

    const unsigned char* set = st.stage(tiles_lds);
    KSTAMP(0)  // staging: raw sub-tile -> LayerNorm -> fp16 pieces -> the two LDS tiles of a set
    st.sync();
    KSTAMP(1)  // waits, barrier, DMA issue
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const unsigned char* tile = set + ch * TILE_BYTES;
      half8 a[4][2];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          a[mb][pl] = *reinterpret_cast<const half8*>(tile + (mb * 16 + l15) * PITCH + kg * 16 + pl * 64);
#define LA_TERM(pl, bexpr)                                                                          \
  _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) _Pragma("unroll") for (int nb = 0; nb < 4; ++nb) \
      acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mb][pl], bexpr, acc[mb][nb], 0, 0, 0);
      if (!LA_ABL(4)) {
      LA_TERM(1, __builtin_bit_cast(half8, wres[ch][nb * 2]))
      LA_TERM(0, __builtin_bit_cast(half8, wres[ch][nb * 2 + 1]))
      LA_TERM(0, __builtin_bit_cast(half8, wres[ch][nb * 2]))
      } else {
        acc[0][0][0] += (float)a[0][0][0] + (float)a[3][1][7];   // (keeps the fragment reads alive)
      }
#undef LA_TERM
    }
    KSTAMP(2)  // fragment reads + 96 projection MFMAs
    const float inv_s = st.inv_scale();

    // ---- k, v of this sub-tile: acc[mb][nb][r] = value(pixel p0 + mb*16 + 4*kg + r, column nb*16 + l15)
    // (1 / block scale and the weight column's 2^-k are both powers of two: one exact multiplier)
    const bool full = p0 + TP <= n;  // uniform: only the last sub-tile of a ragged image masks its pixels
    // (round 3: the column's multiplier k * log2(e) — 1 / block scale, the weight column's 2^-k, log2(e): positive — is
    //  applied inside the exponential's fma; the maximum is taken over the raw accumulators and scaled once)
    float m_new[2], kks[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      kks[db] = inv_s * osc[db] * 1.44269504088896341f;
      float m = -INFINITY;
      if (!full) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (!(p0 + mb * 16 + 4 * kg + r < n)) acc[mb][db][r] = -INFINITY;   // 2^(-inf) = 0 below
      }
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[mb][db][r]);
      m = rows_max(m) * kks[db];
      m_new[db] = fmaxf(m_run[db], m);
    }
    // rescale the running context columns by 2^(m_run - m_new): column d = this lane
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const float fcol = __builtin_amdgcn_exp2f(m_run[db] - m_new[db]);  // 0 on the first sub-tile (m_run = -inf)
      s_run[db] *= fcol;
      ctx[0][db] *= fcol;
      ctx[1][db] *= fcol;
      m_run[db] = m_new[db];
    }
    // ---- ctx[d][e] += sum_n p[n][d] v[n][e] on the fp16 matrix cores (round 2; round 1: 64 fp32 16x16x4 MFMAs per sub-tile,
    // which execute on the vector ALUs).  The accumulator layout is the operand layout: lane (l15, kg) holds column l15
    // (d for p, e for v) of the pixels mb*16 + 4*kg + r, and K slot 8*kg + j of step s is pixel (2s + (j >> 2))*16 + 4*kg
    // + (j & 3) for A and B alike.  p = 2^(k - m) lies in [0, 1]: * 2^10, split; v: one power-of-two scale per wave and
    // sub-tile, split.  v is the A operand: D rows = e, D columns = d = l15 (the transposed context, see above).
    half8 p1[2][2], p2[2][2], v1[2][2], v2[2][2];  // [block][K step]
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float arg = fmaf(acc[2 * ks + (j >> 2)][db][j & 3], kks[db], -m_new[db]);
          pv[j] = LA_ABL(1) ? arg : __builtin_amdgcn_exp2f(arg);  // 2^-inf = 0 beyond n
          s += pv[j];
        }
        la_split8(pv, 1024.f, p1[db][ks], p2[db][ks], LA_ABL(0));
      }
      s = rows_sum(s);
      s_run[db] += s;
    }
    // v: one power-of-two scale for the wave's 64 pixels x 32 columns (fp16 pieces are floating point: a block-wide
    // scale costs range, not precision — as in the conv kernels), so the product is unscaled by one uniform factor
    KSTAMP(3)  // k: maxima, exponentials, sums, rescale of the running context, split of p
    // (round 3: 1 / block scale and the weight column's 2^-k — powers of two — ride on the split's multiplier, which is exact;
    //  the block maximum is taken over the raw accumulators, one multiply per column.  The split's inline asm reads MFMA
    //  results directly: its multiplier depends on the maximum over ALL of them, so every accumulator register has been read
    //  by a compiler-visible instruction, behind hipcc's own hazard padding, before the asm can start)
    unsigned mx = 0u;
    float kv2[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb) {
      // (pixels beyond n were staged as zeros, so their v is exactly 0 without a mask)
      kv2[eb] = inv_s * osc[2 + eb];
      unsigned mr = 0u;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) mr = max(mr, absbits(acc[mb][2 + eb][r]));
      mx = max(mx, absbits(__uint_as_float(mr) * kv2[eb]));
    }
    mx = wave_max_u32(mx);
    const int exv = min(max((int)(mx >> 23), 32), 254);
    const float scv = __uint_as_float((unsigned)(268 - exv) << 23);         // block maximum * scv in [2^14, 2^15)
    const float inv_v = __uint_as_float((unsigned)(exv - 14 - 10) << 23);   // 1 / scv, and the 2^10 of p
#pragma unroll
    for (int eb = 0; eb < 2; ++eb) {
      const float sv = scv * kv2[eb];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float vv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) vv[j] = acc[2 * ks + (j >> 2)][2 + eb][j & 3];
        la_split8(vv, sv, v1[eb][ks], v2[eb][ks], LA_ABL(0));
      }
    }
    KSTAMP(4)  // v: block maximum, split
    float4v t[2][2];  // [eb][db]: four independent chains, term by term; A = v (rows e), B = p (columns d)
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int db = 0; db < 2; ++db)
        t[eb][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v2[eb][0], p1[db][0], float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int db = 0; db < 2; ++db) t[eb][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v2[eb][1], p1[db][1], t[eb][db], 0, 0, 0);
#pragma unroll
    for (int st2 = 0; st2 < 2; ++st2)
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
#pragma unroll
        for (int db = 0; db < 2; ++db)
          t[eb][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1[eb][st2], p2[db][st2], t[eb][db], 0, 0, 0);
#pragma unroll
    for (int st2 = 0; st2 < 2; ++st2)
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
#pragma unroll
        for (int db = 0; db < 2; ++db)
          t[eb][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1[eb][st2], p1[db][st2], t[eb][db], 0, 0, 0);
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int db = 0; db < 2; ++db) ctx[eb][db] += t[eb][db] * inv_v;
    KSTAMP(5)  // 24 context MFMAs + accumulation
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no LDS-DMA may still be on its way when the workgroup ends)
  float* out = partial + ((size_t)(b * nsplit + sp) * 4 + h) * LA_PART;
  if (kg == 0) {
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      out[db * 16 + l15] = m_run[db] * 0.693147180559945309f;   // back to the natural domain the merge works in
      out[32 + db * 16 + l15] = s_run[db];
    }
  }
  // ctx[d][e] row-major: this lane's four consecutive e of column d are one 16 B piece
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
      st4(out + 64 + (db * 16 + l15) * 32 + eb * 16 + 4 * kg,
          make_float4(ctx[eb][db][0], ctx[eb][db][1], ctx[eb][db][2], ctx[eb][db][3]));
#ifdef DMH_STAMPS
  if (lane == 0)
    for (int i = 0; i < 6; ++i) out[i] = (float)tk[i];
#endif
}
#undef KSTAMP

// ------------------------------------------------------------------------------------------ pass 2
// grid = B * nblk; workgroup covers `tiles` sub-tiles; wave = head.
// wq: [head 4][chunk][db 2][plane 2][lane 64] x 16 B (row d = db*16 + (lane & 15)); oscale 128 floats per channel h*32+d
//
// FUSE (C == 64): the rest of the block too — y = to_out(out) + bias (an f16x3 GEMM over the 32 channels of the head,
// the out^T accumulators being its B operand; the four heads' partial sums meet in LDS), LayerNorm over the 64
// channels, + x — so neither `out` nor the to_out result ever reaches HBM: CFG:254-256 (to_out), :103 (Residual).
struct FuseOut {
  const uint4* wo;      // [head 4][c block 4][plane 2][lane 64] x 16 B: to_out weight, K slot (kg, j) = e (j<4 ? 4kg+j : 16+4kg+j-4)
  const float* osc_o;   // 64: 2^-k of the to_out rows
  const float* bias;    // 64
  const float* g_out;   // 64: gain of the LayerNorm after to_out
  float* y;             // [B][n][64]
  float eps;
};
constexpr int YP = 68;                                  // pitch (floats) of a pixel's 64 channels in the exchange buffer
constexpr int YX_BYTES = 4 * TP * YP * 4;               // [head][pixel][channel] partial sums: 69632 B

// DMH_HX_SLOT (never the product: `make hx`, tools/experiments/hazard_hunt/) rebuilds round 2's open case: a four-float LDS
// slot holding 1.0, written once at kernel start, read after the exchange barrier of every sub-tile and multiplied into the
// per-thread to_out scales — a change that cannot alter a result and did, for a few pixels of some launches under load.
#if defined(DMH_HX_SLOT) && defined(DMH_HX_LB1)
#define DMH_QO_WAVES 1
#else
#define DMH_QO_WAVES 2
#endif
template <bool FUSE>
__global__ __launch_bounds__(256, DMH_QO_WAVES) void linattn_qo_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                            const float* __restrict__ g, const uint4* __restrict__ wq,
                                                            const float* __restrict__ oscale, const float* __restrict__ ctxm,
                                                            float* __restrict__ out, int n, int C, int nblk, int tiles,
                                                            float scale, FuseOut fo, const int32_t* __restrict__ rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NBUF = FUSE ? 1 : 2;
  unsigned char* tiles_lds = smem;
  float* yx = reinterpret_cast<float*>(smem + TILE_BYTES);  // FUSE only
#ifdef DMH_HX_SLOT
  float* hx_slot = reinterpret_cast<float*>(smem + TILE_BYTES + YX_BYTES);
  if (FUSE && threadIdx.x == 0) st4(hx_slot, make_float4(1.f, 1.f, 1.f, 1.f));   // (the first staging barrier publishes it)
#endif

  const int jb = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  if (rows && jb >= rows[0]) return;   // (a row subset, common.h)
  const int b = dmh_rows_phys(rows, jb);
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  const int nch = C / KC;

  LA_ABL_LOAD();
  Stager<NBUF> st;
  st.abl = la_abl;
  st.xb = x + (size_t)b * n * C;
  st.g = g;
  st.C = C;
  st.n = n;
  st.c4 = tid & 7;
  st.pix0 = tid >> 3;
  st.init_scale();
  const float* stats_b = stats + (size_t)b * n * 2;

  const uint4* wb = wq + (size_t)h * nch * (4 * 64) + lane;
  // per-row (d = db*16 + 4*kg + r) weight unscale, and the A operand of out^T = ctx^T q'^T as block-scaled fp16 pieces
  // (round 2; round 1 ran this product as 64 fp32 16x16x4 MFMAs per sub-tile, which execute on the vector ALUs):
  // A[m = e][K slot 8*kg + j] = ctx[d(kg, j)][e = eb*16 + l15] with d(kg, j) = j < 4 ? 4*kg + j : 16 + 4*kg + (j - 4) — the
  // order in which a lane of the q^T accumulators holds its 8 values of a pixel column, so q' needs no lane movement.
  // The context of a head is fixed for the workgroup: one block maximum over its 1024 values, split once.
  float osc[2][4];
  half8 c1[2], c2[2];
  float inv_c, mo = 1.f, inv_o = 1.f;
  {
    const float* cb = ctxm + (size_t)(b * 4 + h) * 1024;
    float cv[2][8];
    unsigned mx = 0u;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = db * 16 + 4 * kg + r;
        osc[db][r] = oscale[h * 32 + d];
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
          cv[eb][db * 4 + r] = cb[d * 32 + eb * 16 + l15];
          mx = max(mx, absbits(cv[eb][db * 4 + r]));
        }
      }
    mx = wave_max_u32(mx);
    const int ex = min(max((int)(mx >> 23), 32), 254);
    const float scc = __uint_as_float((unsigned)(268 - ex) << 23);
    inv_c = __uint_as_float((unsigned)(ex - 14 - 17) << 23);  // 1 / scc, and the 2^17 q' carries (below)
#pragma unroll
    for (int eb = 0; eb < 2; ++eb) dmh_split8(cv[eb], scc, c1[eb], c2[eb]);
    if (FUSE) {
      // STATIC block scale of the attention output (round 3; round 2 tried it, c44f485, and took it back, 25ebbcb, because it
      // fed MFMA results straight into the split's inline asm — see the wait states in front of the split below):
      // |o[e]| = |sum_d ctx[d][e] q'[d]| <= max|ctx| * sum_d q'[d] = max|ctx| * scale, so the scale is known per workgroup;
      // taken over ALL FOUR heads it is the same for every wave, and its inverse rides on the per-channel to_out unscale
      // where the heads are summed: no per-pixel maximum (32 and/max + a cross-lane reduction per 16 pixels), no
      // per-accumulator multiply before the split (32) or after the to_out product (64) any more.
      unsigned* xm = reinterpret_cast<unsigned*>(yx);
      if (lane == 0) xm[h] = mx;
      __syncthreads();
      const unsigned ma = max(max(xm[0], xm[1]), max(xm[2], xm[3]));
      const float bo = __uint_as_float(ma) * scale;
      const int exo = min(max((int)(__float_as_uint(bo) >> 23) + 1, 16), 254);   // + 1: sum q' = scale (1 + rounding)
      mo = inv_c * __uint_as_float((unsigned)(268 - exo) << 23);                 // accumulator units -> o * 2^(141 - exo)
      inv_o = __uint_as_float((unsigned)(exo - 14) << 23);
    }
  }
  float* ob = out + (size_t)b * n * 128 + h * 32;
  // FUSE: to_out weight fragments of this head (A operand: rows c, K = the head's 32 channels), kept in registers
  half8 wo1[4], wo2[4];
  if (FUSE) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      wo1[cb] = __builtin_bit_cast(half8, fo.wo[((h * 4 + cb) * 2 + 0) * 64 + lane]);
      wo2[cb] = __builtin_bit_cast(half8, fo.wo[((h * 4 + cb) * 2 + 1) * 64 + lane]);
    }
  }

  if (blk * tiles * TP < n) {
    st.begin_tile(stats_b, blk * tiles * TP);
    st.issue(0);
  }
  uint4 wres[2][4];
  if (FUSE) {
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int i = 0; i < 4; ++i) wres[ch][i] = wb[(size_t)(ch * 4 + i) * 64];
  }
#ifdef DMH_STAMPS
  unsigned long long tk[6] = {0, 0, 0, 0, 0, 0}, t_prev, t_now;
#define LSTAMP(i)                                                                \
  __builtin_amdgcn_sched_barrier(0);                                             \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_now)::"memory"); \
  __builtin_amdgcn_sched_barrier(0);                                             \
  tk[i] += t_now - t_prev;                                                       \
  t_prev = t_now;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
#else
#define LSTAMP(i)
#endif
  for (int tI = 0; tI < tiles; ++tI) {
    const int p0 = (blk * tiles + tI) * TP;
    if (p0 >= n) break;
    float4v acc[2][4];  // q^T: rows d (db, 4*kg + r), columns = pixels (nbn*16 + l15)
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int nbn = 0; nbn < 4; ++nbn) acc[db][nbn] = float4v{0.f, 0.f, 0.f, 0.f};

    auto chunk = [&](int ch, const uint4 (&aq)[4]) __attribute__((always_inline)) {
      const unsigned char* tile = st.stage(tiles_lds);
      if (ch + 1 < nch) st.issue(ch + 1);
      half8 xb[4][2];
#pragma unroll
      for (int nbn = 0; nbn < 4; ++nbn)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          xb[nbn][pl] = *reinterpret_cast<const half8*>(tile + (nbn * 16 + l15) * PITCH + kg * 16 + pl * 64);
#define LA_TERM(aexpr, pl)                                                                           \
  _Pragma("unroll") for (int db = 0; db < 2; ++db) _Pragma("unroll") for (int nbn = 0; nbn < 4; ++nbn) \
      acc[db][nbn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aexpr, xb[nbn][pl], acc[db][nbn], 0, 0, 0);
      if (!LA_ABL(4)) {
      LA_TERM(__builtin_bit_cast(half8, aq[db * 2]), 1)
      LA_TERM(__builtin_bit_cast(half8, aq[db * 2 + 1]), 0)
      LA_TERM(__builtin_bit_cast(half8, aq[db * 2]), 0)
      } else {
        acc[0][0][0] += (float)xb[0][0][0] + (float)xb[3][1][7];   // (keeps the fragment reads alive)
      }
#undef LA_TERM
    };
    if (FUSE) {  // C == 64: both chunks' q fragments stay in registers for the whole workgroup
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) chunk(ch, wres[ch]);
    } else {
      for (int ch = 0; ch < nch; ++ch) {
        uint4 aq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) aq[i] = wb[(size_t)(ch * 4 + i) * 64];
        chunk(ch, aq);
      }
    }
    const float inv_s = st.inv_scale();
    float qsc[2][4];   // 1 / block scale, the weight row's 2^-k and log2(e) in one multiplier (the block scale is static)
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 4; ++r) qsc[db][r] = inv_s * osc[db][r] * 1.44269504088896341f;
    if (tI + 1 < tiles && p0 + TP < n) {
      st.begin_tile(stats_b, p0 + TP);
      st.issue(0);
    }

    LSTAMP(0)  // staging + q projection
    // ---- q' = softmax over the 32 d of each pixel column, * scale
    float rsv[4];
#pragma unroll
    for (int nbn = 0; nbn < 4; ++nbn) {
      float m = -INFINITY;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float qv = acc[db][nbn][r] * qsc[db][r];   // q * log2(e): the softmax runs in the base-2 domain
          acc[db][nbn][r] = qv;
          m = fmaxf(m, qv);
        }
      m = rows_max(m);
      float s = 0.f;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = LA_ABL(1) ? acc[db][nbn][r] - m : __builtin_amdgcn_exp2f(acc[db][nbn][r] - m);  // v_exp_f32: ~1e-7 relative, as in pass 1
          acc[db][nbn][r] = e;
          s += e;
        }
      s = rows_sum(s);
      // one division per pixel column instead of 32 (softmax * scale, CFG:262-263).  * 2^17 (undone in inv_c): q' <= scale
      // < 2^-2 becomes <= 2^15, so the second fp16 piece of every q' that matters is a normal fp16 number
      rsv[nbn] = scale * 131072.f * __builtin_amdgcn_rcpf(s);   // v_rcp_f32 (1 ulp) instead of the ten-instruction IEEE division;
                                                                // the split below multiplies: q' = e * rs as two fp16 pieces
    }
    LSTAMP(1)  // softmax
    // ---- out^T[e][n] = sum_d ctx[d][e] q'[n][d]
    float4v yacc[4][4];  // FUSE: this head's part of to_out: rows c (cb, 4*kg + r), columns = pixels
    // q' of pixel column l15 (0 <= q' * 2^17 <= 2^15) as fp16 pieces: the lane's 8 values are one K = 32 B-fragment slice
    half8 q1[4], q2[4];
#pragma unroll
    for (int nbn = 0; nbn < 4; ++nbn) {
      float e8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e8[j] = acc[j >> 2][nbn][j & 3];
      la_split8(e8, rsv[nbn], q1[nbn], q2[nbn], LA_ABL(0));
    }
    // eight independent accumulation chains, term by term (a chain's MFMAs are eight instructions apart)
    float4v ot[4][2];
#pragma unroll
    for (int nbn = 0; nbn < 4; ++nbn)
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
        ot[nbn][eb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1[eb], q2[nbn], float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
    for (int nbn = 0; nbn < 4; ++nbn)
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
        ot[nbn][eb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(c2[eb], q1[nbn], ot[nbn][eb], 0, 0, 0);
#pragma unroll
    for (int nbn = 0; nbn < 4; ++nbn)
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
        ot[nbn][eb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1[eb], q1[nbn], ot[nbn][eb], 0, 0, 0);
    if (FUSE) {
      // The split reads the MFMA results through inline asm and writes registers hipcc is free to take from MFMA operands
      // that have just died — its hazard recognizer pads neither side of an asm statement.  So: 8 wait states behind the
      // last v_mfma_f32_16x16x32_f16 (4 passes), pinned to every accumulator register the splits read; then ALL four
      // splits; then the 48 to_out MFMAs with no asm statement between them (tests/test_isa_hazards.py scans the result;
      // split-by-split, the second split's destinations were the third-last MFMA's SrcC: round 2's fault, commit 25ebbcb)
      asm volatile("s_nop 7"
                   : "+v"(ot[0][0]), "+v"(ot[0][1]), "+v"(ot[1][0]), "+v"(ot[1][1]), "+v"(ot[2][0]), "+v"(ot[2][1]),
                     "+v"(ot[3][0]), "+v"(ot[3][1]));
      half8 h1[4], h2[4];
#pragma unroll
      for (int nbn = 0; nbn < 4; ++nbn) {
        // the 8 values of this lane (e = eb*16 + 4*kg + r of pixel column l15) are one K = 32 B-fragment slice, split under
        // the workgroup's static scale (see the head of the kernel)
        float o8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = ot[nbn][j >> 2][j & 3];
        la_split8(o8, mo, h1[nbn], h2[nbn], LA_ABL(0));
      }
#pragma unroll
      for (int nbn = 0; nbn < 4; ++nbn)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) yacc[cb][nbn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wo1[cb], h2[nbn], float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int nbn = 0; nbn < 4; ++nbn)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) yacc[cb][nbn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wo2[cb], h1[nbn], yacc[cb][nbn], 0, 0, 0);
#pragma unroll
      for (int nbn = 0; nbn < 4; ++nbn)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) yacc[cb][nbn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wo1[cb], h1[nbn], yacc[cb][nbn], 0, 0, 0);
    } else {
#pragma unroll
      for (int nbn = 0; nbn < 4; ++nbn) {
        const int pix = p0 + nbn * 16 + l15;  // column = pixel; rows e = eb*16 + 4*kg + r: four consecutive channels
        if (pix < n) {
#pragma unroll
          for (int eb = 0; eb < 2; ++eb) {
            const float4v o = ot[nbn][eb] * inv_c;
            st4(ob + (size_t)pix * 128 + eb * 16 + 4 * kg, make_float4(o[0], o[1], o[2], o[3]));
          }
        }
      }
    }
    LSTAMP(2)  // q' split, ctx product, to_out product
    if (FUSE) {
      // partial sums of the four heads -> LDS [head][pixel][channel]; lane (pixel column l15 of block nbn) holds the four
      // consecutive channels cb*16 + 4*kg .. +3
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int nbn = 0; nbn < 4; ++nbn)
          st4(yx + ((h * TP) + nbn * 16 + l15) * YP + cb * 16 + 4 * kg,
              make_float4(yacc[cb][nbn][0], yacc[cb][nbn][1], yacc[cb][nbn][2], yacc[cb][nbn][3]));
      // the residual rows of this thread's four pixels (L2: the workgroup staged them a moment ago) are requested before the
      // barrier, together: loaded one by one where they are added, each exposed its round trip
      const int quad = tid & 15;
      float4 xr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pix = LA_ABL(3) ? (tid >> 4) : min(p0 + (tid >> 4) + 16 * i, n - 1);
        xr[i] = ld4(st.xb + (size_t)pix * 64 + quad * 4);
      }
      float4 oq = ld4(fo.osc_o + quad * 4);
      const float4 bq4 = ld4(fo.bias + quad * 4), gq4 = ld4(fo.g_out + quad * 4);
      oq.x *= inv_o;   // 2^-k of the to_out row and 1 / (static output scale): powers of two
      oq.y *= inv_o;
      oq.z *= inv_o;
      oq.w *= inv_o;
      __syncthreads();
      LSTAMP(3)  // exchange write + barrier
#ifdef DMH_HX_SLOT
#ifdef DMH_HX_CONST
      const float4 hx = make_float4(1.f, 1.f, 1.f, 1.f);
#else
      const float4 hx = ld4(hx_slot);
#endif
      const float4 oqx = make_float4(oq.x * hx.x, oq.y * hx.y, oq.z * hx.z, oq.w * hx.w);
#define DMH_OQ oqx
#else
#define DMH_OQ oq
#endif
      // 64 pixels x 16 channel quads: sum the heads, undo the weight scale, + bias, LayerNorm over the 64 channels (two
      // passes, as chan_layernorm_kernel), * g, + x
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (LA_ABL(2)) break;
        const int pl = (tid >> 4) + 16 * i;  // pixel of the sub-tile
        const float* yp = yx + pl * YP + quad * 4;
        const float4 y0 = ld4(yp), y1 = ld4(yp + TP * YP), y2 = ld4(yp + 2 * TP * YP), y3 = ld4(yp + 3 * TP * YP);
        float4 v;
        v.x = fmaf((y0.x + y1.x) + (y2.x + y3.x), DMH_OQ.x, bq4.x);
        v.y = fmaf((y0.y + y1.y) + (y2.y + y3.y), DMH_OQ.y, bq4.y);
        v.z = fmaf((y0.z + y1.z) + (y2.z + y3.z), DMH_OQ.z, bq4.z);
        v.w = fmaf((y0.w + y1.w) + (y2.w + y3.w), DMH_OQ.w, bq4.w);
        float sm = (v.x + v.y) + (v.z + v.w);
        sm = row16_sum(sm);
        const float mean = sm / 64.f;
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        float qs = (dx * dx + dy * dy) + (dz * dz + dw * dw);
        qs = row16_sum(qs);
        const float rs = __builtin_amdgcn_rsqf(qs * (1.f / 64.f) + fo.eps);   // v_rsq_f32 (1 ulp) instead of sqrt + IEEE division
        const int pix = p0 + pl;
        if (pix < n) {
          float4 r4;
          r4.x = dx * rs * gq4.x + xr[i].x;
          r4.y = dy * rs * gq4.y + xr[i].y;
          r4.z = dz * rs * gq4.z + xr[i].z;
          r4.w = dw * rs * gq4.w + xr[i].w;
          st4(fo.y + ((size_t)b * n + pix) * 64 + quad * 4, r4);
        }
      }
    }
    LSTAMP(4)  // LayerNorm + residual + store
    if (FUSE) __syncthreads();  // the exchange buffer is free for the next sub-tile
    LSTAMP(5)  // end barrier
  }
#ifdef DMH_STAMPS
  if (FUSE && lane == 0) {   // diagnostic build only: the counters overwrite the first outputs of the workgroup's first sub-tile
    float* d = fo.y + ((size_t)b * n + (size_t)blk * tiles * TP) * 64 + h * 8;
    for (int i = 0; i < 6; ++i) d[i] = (float)tk[i];
  }
#endif
}
#undef LSTAMP

// ------------------------------------------------------------------------------------------ weight packing
// to_qkv weight [384][C] (1x1, no bias): rows 0..127 q, 128..255 k, 256..383 v, row = part*128 + head*32 + d.
// per-row scale 2^k with max |w| * 2^k in [2^14, 2^15); oscale = 2^-k
__global__ __launch_bounds__(64) void linattn_wscale_kernel(const float* __restrict__ w, float* __restrict__ osc_q,
                                                            float* __restrict__ osc_kv, int C) {
  const int row = blockIdx.x;  // 0..383
  float m = 0.f;
  for (int i = threadIdx.x; i < C; i += 64) m = fmaxf(m, fabsf(w[(size_t)row * C + i]));
  for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if (threadIdx.x == 0) {
    float s = 1.f;
    if (m > 0.f && m < 3.0e38f) {
      int e;
      frexpf(m, &e);
      s = ldexpf(1.f, min(max(e - 15, -100), 100));
    }
    if (row < 128) {
      osc_q[row] = s;
    } else {
      // kv order: (head, nb: k lo, k hi, v lo, v hi, col)
      const int part = row / 128 - 1, hh = (row % 128) / 32, d = row % 32;
      osc_kv[(hh * 4 + part * 2 + d / 16) * 16 + d % 16] = s;
    }
  }
}

// kv: fp16 index ((((h * nch + ch) * 4 + nb) * 2 + plane) * 64 + lane) * 8 + j ; q: ((((h * nch + ch) * 2 + db) * 2 + plane) * 64 + lane) * 8 + j
__global__ void linattn_pack_kernel(const float* __restrict__ w, const float* __restrict__ osc_q,
                                    const float* __restrict__ osc_kv, _Float16* __restrict__ pq,
                                    _Float16* __restrict__ pkv, int C, int64_t total_q, int64_t total_kv) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nch = C / KC;
  if (idx < total_q) {
    int64_t r = idx;
    const int j = r % 8;
    r /= 8;
    const int lane = r % 64;
    r /= 64;
    const int plane = r % 2;
    r /= 2;
    const int db = r % 2;
    r /= 2;
    const int ch = r % nch;
    const int hh = r / nch;
    const int row = hh * 32 + db * 16 + (lane & 15);
    const int c = ch * KC + (lane >> 4) * 8 + j;
    const float ws = w[(size_t)row * C + c] / osc_q[row];
    const _Float16 g1 = (_Float16)ws;
    pq[idx] = plane == 0 ? g1 : (_Float16)(ws - (float)g1);
  }
  if (idx < total_kv) {
    int64_t r = idx;
    const int j = r % 8;
    r /= 8;
    const int lane = r % 64;
    r /= 64;
    const int plane = r % 2;
    r /= 2;
    const int nb = r % 4;
    r /= 4;
    const int ch = r % nch;
    const int hh = r / nch;
    const int row = 128 + (nb >> 1) * 128 + hh * 32 + (nb & 1) * 16 + (lane & 15);
    const int c = ch * KC + (lane >> 4) * 8 + j;
    const float ws = w[(size_t)row * C + c] / osc_kv[(hh * 4 + nb) * 16 + (lane & 15)];
    const _Float16 g1 = (_Float16)ws;
    pkv[idx] = plane == 0 ? g1 : (_Float16)(ws - (float)g1);
  }
}

// ------------------------------------------------------------------------------------------ C ABI
// packed image (floats): [q fragments: C*128 fp16 * 2 planes][kv fragments: C*256 fp16 * 2 planes][osc_q 128][osc_kv 256]
extern "C" int64_t dmh_linattn_fused_pack_floats(int C) { if (!dmh_dims_ok({C})) return -1; return (int64_t)C * 128 + (int64_t)C * 256 + 128 + 256; }

extern "C" int dmh_linattn_fused_pack(const float* w_qkv, float* wpack, int C, void* stream) {
  DMH_REQUIRE(w_qkv && wpack && C > 0 && C % KC == 0, "dmh_linattn_fused_pack: C must be a multiple of 32 (got %d)", C);
  hipStream_t s = (hipStream_t)stream;
  float* osc_q = wpack + (int64_t)C * 384;
  float* osc_kv = osc_q + 128;
  hipLaunchKernelGGL(linattn_wscale_kernel, dim3(384), dim3(64), 0, s, w_qkv, osc_q, osc_kv, C);
  const int64_t total_q = (int64_t)C * 128 * 2, total_kv = (int64_t)C * 256 * 2;
  hipLaunchKernelGGL(linattn_pack_kernel, dim3((unsigned)cdiv64(total_kv, 256)), dim3(256), 0, s, w_qkv, osc_q, osc_kv,
                     reinterpret_cast<_Float16*>(wpack), reinterpret_cast<_Float16*>(wpack + (int64_t)C * 128), C,
                     total_q, total_kv);
  DMH_CHECK_LAUNCH("dmh_linattn_fused_pack");
  return DMH_OK;
}

static int fused_tiles(int B, int n) {
  // sub-tiles per workgroup: a function of n ONLY — the split of a sample's pixels fixes the order of its online-softmax
  // merges, and a sample's result must not depend on how many other samples share the launch (rows stay bitwise
  // independent: what makes sharding across GPUs and the two-stream CFG split exact).
  (void)B;
  const int nt = cdiv(n, TP);
  const int t = nt / 32;  // (round 2: 32 splits per sample at most — 64 in round 1; swept 8 .. 64: +0.4 % images/s, 64x64 levels -9 %)
  return t < 1 ? 1 : (t > 8 ? 8 : t);
}

extern "C" int dmh_linattn_fused_splits(int B, int n) {
  if (!dmh_dims_ok({B}) || !dmh_dims_ok({n}, 1, 1 << 26)) return -1;
  return cdiv(cdiv(n, TP), fused_tiles(B, n));
}

// pass 1: partial[B][splits][4][LA_PART]  (then dmh_linattn_merge with the same split count)
extern "C" int dmh_linattn_fused_context(const float* x, const float* stats, const float* ln_g, const float* wpack,
                                         float* partial, int B, int n, int C, const int32_t* rows, void* stream) {
  DMH_REQUIRE(x && stats && ln_g && wpack && partial, "dmh_linattn_fused_context: null pointer");
  DMH_REQUIRE(B > 0 && n > 0 && C > 0 && C % KC == 0, "dmh_linattn_fused_context: bad shape (C=%d)", C);
  const int tiles = fused_tiles(B, n), nsplit = cdiv(cdiv(n, TP), tiles);
  const uint4* wkv = reinterpret_cast<const uint4*>(wpack + (int64_t)C * 128);
  const float* osc_kv = wpack + (int64_t)C * 384 + 128;
  static const int use_ring = [] {   // development knob for same-box A/Bs: DMH_LA_RING=0 -> round 2's register staging
    const char* e = getenv("DMH_LA_RING");
    return e ? atoi(e) : 1;
  }();
  if (C == 2 * KC && use_ring) {
    constexpr int LDS = 4 * TILE_BYTES + RingStager::BYTES;
    static bool attr = false;
    if (!attr) {
      hipError_t e = hipFuncSetAttribute((const void*)linattn_kv_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      DMH_REQUIRE(e == hipSuccess, "dmh_linattn_fused_context: cannot raise the LDS limit");
      attr = true;
    }
    hipLaunchKernelGGL(linattn_kv_ring_kernel, dim3(B * nsplit), dim3(256), LDS, (hipStream_t)stream, x, stats, ln_g, wkv,
                       osc_kv, partial, n, C, nsplit, tiles, rows);
  } else if (C == 2 * KC)
    hipLaunchKernelGGL(linattn_kv_kernel<2>, dim3(B * nsplit), dim3(256), 2 * TILE_BYTES, (hipStream_t)stream, x, stats, ln_g,
                       wkv, osc_kv, partial, n, C, nsplit, tiles, rows);
  else
    hipLaunchKernelGGL(linattn_kv_kernel<0>, dim3(B * nsplit), dim3(256), 2 * TILE_BYTES, (hipStream_t)stream, x, stats, ln_g,
                       wkv, osc_kv, partial, n, C, nsplit, tiles, rows);
  DMH_CHECK_LAUNCH("dmh_linattn_fused_context");
  return DMH_OK;
}

// pass 2: out[B][n][128]
extern "C" int dmh_linattn_fused_apply(const float* x, const float* stats, const float* ln_g, const float* wpack,
                                       const float* ctx, float* out, int B, int n, int C, float scale, const int32_t* rows,
                                       void* stream) {
  DMH_REQUIRE(x && stats && ln_g && wpack && ctx && out, "dmh_linattn_fused_apply: null pointer");
  DMH_REQUIRE(B > 0 && n > 0 && C > 0 && C % KC == 0, "dmh_linattn_fused_apply: bad shape (C=%d)", C);
  const int tiles = fused_tiles(B, n), nblk = cdiv(cdiv(n, TP), tiles);
  const uint4* wq = reinterpret_cast<const uint4*>(wpack);
  const float* osc_q = wpack + (int64_t)C * 384;
  FuseOut fo = {};
  hipLaunchKernelGGL(linattn_qo_kernel<false>, dim3(B * nblk), dim3(256), 2 * TILE_BYTES, (hipStream_t)stream, x, stats,
                     ln_g, wq, osc_q, ctx, out, n, C, nblk, tiles, scale, fo, rows);
  DMH_CHECK_LAUNCH("dmh_linattn_fused_apply");
  return DMH_OK;
}

// ---- to_out weight [64][128] (1x1) for the fully fused pass 2 (C == 64)
// fp16 index (((h * 4 + cb) * 2 + plane) * 64 + lane) * 8 + j  ->  piece of w[c = cb*16 + (lane & 15)][h*32 + e],
// e = j < 4 ? 4*kg + j : 16 + 4*kg + (j - 4), kg = lane >> 4; then 64 floats 2^-k per row c
__global__ __launch_bounds__(64) void linattn_wo_scale_kernel(const float* __restrict__ w, float* __restrict__ osc) {
  const int c = blockIdx.x;
  float m = fmaxf(fabsf(w[c * 128 + threadIdx.x]), fabsf(w[c * 128 + 64 + threadIdx.x]));
  for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if (threadIdx.x == 0) {
    float s = 1.f;
    if (m > 0.f && m < 3.0e38f) {
      int e;
      frexpf(m, &e);
      s = ldexpf(1.f, min(max(e - 15, -100), 100));
    }
    osc[c] = s;
  }
}
__global__ void linattn_wo_pack_kernel(const float* __restrict__ w, const float* __restrict__ osc,
                                       _Float16* __restrict__ wp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // 4*4*2*64*8 = 16384
  if (idx >= 16384) return;
  int r = idx;
  const int j = r % 8;
  r /= 8;
  const int lane = r % 64;
  r /= 64;
  const int plane = r % 2;
  r /= 2;
  const int cb = r % 4;
  const int hh = r / 4;
  const int c = cb * 16 + (lane & 15), kg = lane >> 4;
  const int e = j < 4 ? 4 * kg + j : 16 + 4 * kg + (j - 4);
  const float ws = w[c * 128 + hh * 32 + e] / osc[c];
  const _Float16 g1 = (_Float16)ws;
  wp[idx] = plane == 0 ? g1 : (_Float16)(ws - (float)g1);
}

extern "C" int64_t dmh_linattn_out_pack_floats(void) { return 16384 / 2 + 64; }

extern "C" int dmh_linattn_out_pack(const float* w_out, float* wpack, void* stream) {
  DMH_REQUIRE(w_out && wpack, "dmh_linattn_out_pack: null pointer");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(linattn_wo_scale_kernel, dim3(64), dim3(64), 0, s, w_out, wpack + 8192);
  hipLaunchKernelGGL(linattn_wo_pack_kernel, dim3(64), dim3(256), 0, s, w_out, wpack + 8192,
                     reinterpret_cast<_Float16*>(wpack));
  DMH_CHECK_LAUNCH("dmh_linattn_out_pack");
  return DMH_OK;
}

// pass 2 with to_out + bias, LayerNorm, + x fused in (C == 64): y[B][n][64] = x + LN(to_out(attention(LN(x))))
extern "C" int dmh_linattn_fused_apply_out(const float* x, const float* stats, const float* ln_g, const float* wpack,
                                           const float* ctx, const float* wopack, const float* out_bias,
                                           const float* out_ln_g, float* y, int B, int n, int C, float scale,
                                           float eps, const int32_t* rows, void* stream) {
  DMH_REQUIRE(x && stats && ln_g && wpack && ctx && wopack && out_bias && out_ln_g && y,
              "dmh_linattn_fused_apply_out: null pointer");
  DMH_REQUIRE(B > 0 && n > 0 && C == 64, "dmh_linattn_fused_apply_out: only C == 64 (got %d)", C);
  const int tiles = fused_tiles(B, n), nblk = cdiv(cdiv(n, TP), tiles);
  const uint4* wq = reinterpret_cast<const uint4*>(wpack);
  const float* osc_q = wpack + (int64_t)C * 384;
  FuseOut fo;
  fo.wo = reinterpret_cast<const uint4*>(wopack);
  fo.osc_o = wopack + 8192;
  fo.bias = out_bias;
  fo.g_out = out_ln_g;
  fo.y = y;
  fo.eps = eps;
#ifdef DMH_HX_SLOT
  constexpr int LDS = TILE_BYTES + YX_BYTES + 16;
#else
  constexpr int LDS = TILE_BYTES + YX_BYTES;
#endif
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)linattn_qo_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    DMH_REQUIRE(e == hipSuccess, "dmh_linattn_fused_apply_out: cannot raise the LDS limit");
    attr = true;
  }
  hipLaunchKernelGGL(linattn_qo_kernel<true>, dim3(B * nblk), dim3(256), LDS, (hipStream_t)stream, x, stats, ln_g, wq,
                     osc_q, ctx, nullptr, n, C, nblk, tiles, scale, fo, rows);
  DMH_CHECK_LAUNCH("dmh_linattn_fused_apply_out");
  return DMH_OK;
}
