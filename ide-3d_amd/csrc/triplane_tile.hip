// triplane_tile.hip — tri-plane gather for samples that come from a ray grid (LDS-staged plane regions).
//
// Same result as `ide3d_triplane_sample` (triplane.hip; dnnlib/util.py:580-617), for the case the ray-marcher
// produces: coords laid out [image][ray row][ray column][depth step].  The caller passes that shape as a hint; the
// hint only decides how samples are grouped, never what is computed, so any coordinates give the right answer.
//
// Why: the flat kernel moves 12 taps x C*4 bytes = 1.5 KB per sample through the L1/TA return path (64 B/clk/CU),
// which saturates at ~42 % of the HBM roofline.  Neighbouring rays and consecutive depth steps
// touch the same texels: an 8 x 8 ray tile x 4 depth steps (256 samples, 3072 tap lines) touches only ~500-600
// distinct 128-byte lines.  Per chunk of 256 samples a workgroup
//   A. computes the taps (lane = sample; masked bilinear weights + footprint origin) and reduces the bounding box of
//      the footprints per plane (packed 16-bit min / max on the DPP cross-lane path + one LDS exchange);
//   B. copies the bounding boxes into LDS (`global_load_dwordx4` + `ds_write_b128`, 8 lines per wave instruction, two
//      planes' loads in flight at once) — 1/5 of the bytes the flat kernel pulls through the TA.  Boxes are in
//      *virtual* texel coordinates (index + 1); only boxes that lie inside the plane are staged, so a sample's four taps
//      sit at +0, +1 line, +1 row, +1 row +1 line and need no per-tap address arithmetic;
//   C. blends from LDS (lane = (sample slot, 4-channel slice), `ds_read_b128`) and writes 16-byte output slices with
//      non-temporal stores.
// Planes whose boxes do not fit the LDS budget or touch the plane border (grazing rays, wild coordinates) are read
// with buffer loads for that chunk, exactly like the flat kernel: decided per plane and per chunk, wave-uniform, with
// one blend variant per combination so that LDS and buffer-load code never share registers.
// Two workgroups per CU (80 KB LDS each) overlap one group's fetch latency with the other's blending.
//
// Measured (MI355X, benchmark shape): 89 us = 45 % of the 8 TB/s roofline vs 95 us = 42 % for the
// flat kernel; HBM and TA traffic are no longer the limit — with 80 KB of LDS per workgroup only two waves share a
// SIMD, a wave issues one instruction per 4 cycles, and the kernel needs ~1700 instructions per wave and chunk
// (cycle stamps: build with EXTRA=-DIDE3D_TT_TRACE, scripts/gather_trace.py).
//
// Arithmetic (tap indices, weights, blend order) is identical to triplane.hip: results are bit-equal (tests).
#include "common.h"
#include "knobs.h"
#include "triplane_tap.h"

namespace ide3d {

#ifdef IDE3D_TT_TRACE
// Developer aid (make EXTRA=-DIDE3D_TT_TRACE): cycle stamps of one wave at the phase boundaries of every chunk,
// read back with ide3d_debug_tt().  Not part of the ABI, not built by default.
__device__ unsigned long long g_tt_dbg[256];
#define IDE3D_TS(k) if (blockIdx.x == 300 && threadIdx.x == 0) g_tt_dbg[ch * 8 + (k)] = __builtin_readcyclecounter();
#else
#define IDE3D_TS(k)
#endif

namespace {

constexpr int TT_EDGE = 8;                    // ray tile edge
constexpr int TT_DS = 4;                      // depth steps per chunk
constexpr int TT_C = 32;                      // channels per plane (line = 128 B)
constexpr int TT_LINE = TT_C * 4;             // bytes
constexpr int TT_CAP = 504;                   // LDS lines: 504 * 128 + 256 * 64 + 128 = 81 024 B -> 2 workgroups / CU
constexpr int PC_CAP = 504;                   // producer / consumer kernel, lines per buffer: 2 x (504 x 128 + 16 KB tap table) + 512 B = 162 304 B of the 160 KB

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

struct TileArgs {
    const float* planes;      // first image of the group
    const float* coords;      // first sample of the group
    float* out;
    unsigned sN_bytes;        // image stride
    unsigned group_bytes;     // bytes addressable from `planes` (buffer resource range)
    int sH, sW;               // element strides of the plane rows / pixels
    int H, W;
    int rays_w, rays_per_image, steps;
    int tiles_x, tiles_per_image, segs, chunks_per_seg;
};

// One axis of a bilinear tap: ATen grid_sampler_unnormalize + floor, exactly as make_tap() (triplane_tap.h) does it.
// The three planes share coordinates ((x,y), (y,z), (x,z)), so four axis set-ups serve six plane axes.
struct AxisTap {
    unsigned v;               // virtual index of the low tap: clamp(floor(u), -1, size - 1) + 1  in [0, size]
    float a, b;               // weights of the low / high tap:  (floor(u) + 1) - u,  u - floor(u)
    bool ok0, ok1;            // low / high tap inside the plane and u finite
};

__device__ __forceinline__ AxisTap axis_tap(float c, int size) {
    AxisTap t;
    const float u = unnormalize(c, size);
    const float fu = floorf(u);
    const float fuc = fminf(fmaxf(fu, -2.0f), (float)size + 1.0f);
    const int i = (int)fuc;
    t.a = __fsub_rn(__fadd_rn(fu, 1.0f), u);
    t.b = __fsub_rn(u, fu);
    const bool finite = (u == u) && fu == fuc;
    t.ok0 = finite && i >= 0 && i < size;
    t.ok1 = finite && i + 1 >= 0 && i + 1 < size;
    t.v = (unsigned)(min(max(i, -1), size - 1) + 1);
    return t;
}

struct PlaneTap {             // one plane of one sample
    unsigned vx, vy;          // virtual nw texel
    float w00, w01, w10, w11; // masked weights (0 for out-of-plane taps: zeros padding)
};

__device__ __forceinline__ PlaneTap plane_tap(const AxisTap& x, const AxisTap& y) {
    PlaneTap t;
    t.vx = x.v; t.vy = y.v;
    t.w00 = (x.ok0 && y.ok0) ? __fmul_rn(x.a, y.a) : 0.f;
    t.w01 = (x.ok1 && y.ok0) ? __fmul_rn(x.b, y.a) : 0.f;
    t.w10 = (x.ok0 && y.ok1) ? __fmul_rn(x.a, y.b) : 0.f;
    t.w11 = (x.ok1 && y.ok1) ? __fmul_rn(x.b, y.b) : 0.f;
    return t;
}

struct SampleTaps {
    PlaneTap pl[3];
    unsigned ax[2];           // packed virtual indices: (x on W | y on H << 16), (y on W | z on H << 16)
};

__device__ __forceinline__ SampleTaps sample_taps(float cx, float cy, float cz, int W, int H) {
    const AxisTap xw = axis_tap(cx, W), yh = axis_tap(cy, H), yw = axis_tap(cy, W), zh = axis_tap(cz, H);
    SampleTaps s;
    s.pl[0] = plane_tap(xw, yh); s.pl[1] = plane_tap(yw, zh); s.pl[2] = plane_tap(xw, zh);
    s.ax[0] = xw.v | (yh.v << 16); s.ax[1] = yw.v | (zh.v << 16);
    return s;
}

__device__ __forceinline__ unsigned pk_min(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
// Wave-wide reduction of packed 16-bit pairs on the DPP cross-lane path (no LDS traffic): inclusive row scan
// (row_shr 1, 2, 4, 8), then row_bcast15 / row_bcast31 carry the row totals to lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_mov(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
}
template <bool MAX>
__device__ __forceinline__ unsigned wave_reduce_pk(unsigned v) {
#define IDE3D_RED_STEP(CTRL, RM) { const unsigned o = dpp_mov<CTRL, RM>(v); v = MAX ? pk_max(v, o) : pk_min(v, o); }
    IDE3D_RED_STEP(0x111, 0xf) IDE3D_RED_STEP(0x112, 0xf) IDE3D_RED_STEP(0x114, 0xf) IDE3D_RED_STEP(0x118, 0xf)
    IDE3D_RED_STEP(0x142, 0xa) IDE3D_RED_STEP(0x143, 0xc)
#undef IDE3D_RED_STEP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
// The four reductions a chunk needs (min / max of two packed pairs), step by step in lock-step: four independent dependency chains,
// so the two wait states a DPP read needs after the VALU write of its source are filled with the other chains instead of s_nops.
__device__ __forceinline__ void wave_reduce_pk4(unsigned& lo0, unsigned& lo1, unsigned& hi0, unsigned& hi1) {
#define IDE3D_RED4(CTRL, RM) { const unsigned a = dpp_mov<CTRL, RM>(lo0), b = dpp_mov<CTRL, RM>(lo1), c = dpp_mov<CTRL, RM>(hi0), d = dpp_mov<CTRL, RM>(hi1); \
                               lo0 = pk_min(lo0, a); lo1 = pk_min(lo1, b); hi0 = pk_max(hi0, c); hi1 = pk_max(hi1, d); }
    IDE3D_RED4(0x111, 0xf) IDE3D_RED4(0x112, 0xf) IDE3D_RED4(0x114, 0xf) IDE3D_RED4(0x118, 0xf) IDE3D_RED4(0x142, 0xa) IDE3D_RED4(0x143, 0xc)
#undef IDE3D_RED4
    lo0 = (unsigned)__builtin_amdgcn_readlane((int)lo0, 63); lo1 = (unsigned)__builtin_amdgcn_readlane((int)lo1, 63);
    hi0 = (unsigned)__builtin_amdgcn_readlane((int)hi0, 63); hi1 = (unsigned)__builtin_amdgcn_readlane((int)hi1, 63);
}

// Region of one plane for one chunk (all wave-uniform).
struct Region {
    unsigned x0, y0;          // virtual origin
    unsigned bw, bh;          // lines per row, rows
    unsigned base;            // first LDS line
    bool staged;
};

// ---- C: blending ------------------------------------------------------------------------------------------------------
// Round r of wave `wid`, slot k handles workgroup sample sb = wid*64 + r*8 + k = (tile ray sb >> 2, depth sb & 3); lane =
// (slot, 4-channel slice).  Tap table entry of a sample: [0..2] = the four masked weights of plane pl, [3] = per plane
// the byte offset of the nw line — an LDS offset for staged planes, else a plane offset with "x / y neighbour is another
// texel" in bits 0 / 1.  MASK bit pl = plane pl is staged in LDS (compile time: one variant per combination, so LDS and
// buffer-load code never share registers).
struct TapEntry { u32x4 w[3]; u32x4 offs; };

__device__ __forceinline__ TapEntry load_taps(const u32x4 (*s_tap)[4], unsigned sb) {
    TapEntry t;
    t.offs = s_tap[sb][3]; t.w[0] = s_tap[sb][0]; t.w[1] = s_tap[sb][1]; t.w[2] = s_tap[sb][2];
    return t;
}

// the four tap lines (this lane's 16-byte slice) of one plane of one sample
template <bool STAGED>
__device__ __forceinline__ void load_lines(const TileArgs& p, const unsigned char* s_lines, __amdgpu_buffer_rsrc_t rsrc,
                                           unsigned pitch, unsigned e, unsigned ch_bytes, f32x4_t (&v)[4]) {
    if (STAGED) {
        const unsigned char* l0 = s_lines + (e | ch_bytes);
        const unsigned char* l1 = l0 + pitch;
        v[0] = *reinterpret_cast<const f32x4_t*>(l0);
        v[1] = *reinterpret_cast<const f32x4_t*>(l0 + TT_LINE);
        v[2] = *reinterpret_cast<const f32x4_t*>(l1);
        v[3] = *reinterpret_cast<const f32x4_t*>(l1 + TT_LINE);
    } else {
        const unsigned o = (e & ~3u) + ch_bytes;
        const unsigned dx = (e & 1u) * ((unsigned)p.sW * 4u);
        const unsigned dy = ((e >> 1) & 1u) * ((unsigned)p.sH * 4u);
        v[0] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, 0, 0));
        v[1] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(o + dx), 0, 0));
        v[2] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(o + dy), 0, 0));
        v[3] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(o + dy + dx), 0, 0));
    }
}

// All 8 rounds of a wave for one chunk, fully unrolled.  Results stay in registers and are stored together at the end: on
// gfx9 stores and loads share `vmcnt`, so a store between two rounds would make the next round's buffer loads wait for
// it.  (An explicit software pipeline over (round, plane) units pinned with sched_barriers measured 15 % slower than
// letting the compiler schedule the unrolled rounds; 512-thread workgroups at 128 VGPRs measured 40 % slower.)
template <int MASK, int ROUNDS = 8>     // ROUNDS rounds of 8 samples per wave: 8 (four blending waves) or 4 (eight)
__device__ __forceinline__ void blend_chunk(const TileArgs& p, const unsigned char* s_lines, const u32x4 (*s_tap)[4],
                                               __amdgpu_buffer_rsrc_t rsrc, const unsigned (&pitch)[3],
                                               int wid, int slot, int cl, unsigned ray00, unsigned img, unsigned step0) {
    const unsigned ch_bytes = (unsigned)cl * 16u;
    f32x4_t res[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const unsigned sb = (unsigned)wid * (unsigned)(ROUNDS * 8) + (unsigned)r * 8u + (unsigned)slot;
        const TapEntry t = load_taps(s_tap, sb);
        f32x4_t v[3][4];
        load_lines<(MASK & 1) != 0>(p, s_lines, rsrc, pitch[0], t.offs[0], ch_bytes, v[0]);
        load_lines<(MASK & 2) != 0>(p, s_lines, rsrc, pitch[1], t.offs[1], ch_bytes, v[1]);
        load_lines<(MASK & 4) != 0>(p, s_lines, rsrc, pitch[2], t.offs[2], ch_bytes, v[2]);
        f32x4_t acc[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const f32x4_t w = __builtin_bit_cast(f32x4_t, t.w[pl]);
            f32x4_t a = {0.f, 0.f, 0.f, 0.f};
            a += v[pl][0] * w.x; a += v[pl][1] * w.y; a += v[pl][2] * w.z; a += v[pl][3] * w.w;
            acc[pl] = a;
        }
        res[r] = (acc[0] + acc[1]) + acc[2];
    }
    // tile ray of (wave, round, slot): wid * ROUNDS * 2 + r * 2 + (slot >> 2) = (row, column) of the 8 x 8 tile
    constexpr int RPW = ROUNDS / 4;                 // tile rows per wave
    const unsigned ray_l = ray00 + (unsigned)wid * (unsigned)RPW * (unsigned)p.rays_w + ((unsigned)slot >> 2);
    float* const o_lane = p.out + ((size_t)(img * (unsigned)p.rays_per_image + ray_l) * (unsigned)p.steps + step0 + ((unsigned)slot & 3u)) * TT_C + cl * 4;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const size_t d = (size_t)((unsigned)(r >> 2) * (unsigned)p.rays_w + (unsigned)(r & 3) * 2u) * (unsigned)p.steps * TT_C;
        __builtin_nontemporal_store(res[r], reinterpret_cast<f32x4_t*>(o_lane + d));
    }
}
// ---- B: staging -------------------------------------------------------------------------------------------------------
// LDS line q = ry * bw + rx of a region holds virtual texel (x0 + rx, y0 + ry).  A wave instruction moves the 8 lines
// q0 .. q0+7 (lane = (line, 16-byte slice)); wave w takes the segments w, w+4, ... (at most TT_SEGS: regions are limited
// to TT_SEGS * 32 lines).  Loads go through registers (`global_load_dwordx4` + `ds_write_b128`): measured on MI355X,
// LDS-DMA (`global_load_lds`) with per-lane addresses was no faster and cannot be batched.  All loads of a chunk are
// issued before the first LDS write.  Only regions that lie inside the plane are staged (no clamping), so a lane's
// address advances by a constant per segment plus a constant when its line index wraps into the next region row:
// ~8 VALU per load, no scalar work (the scalar unit is shared by the whole CU and a wave issues one instruction per
// 4 cycles, so instruction count — of any kind — is what bounds this kernel).
constexpr int TT_SEGS_A = 10;                 // register set A: planes 0 and 2 (regions up to 320 lines)
constexpr int TT_SEGS_B = 8;                  // register set B: plane 1 (up to 256 lines)

template <int NSEG>
__device__ __forceinline__ void stage_issue(const TileArgs& p, const Region& R, int pl, unsigned img_bytes,
                                            int wid, int slot, unsigned ch_bytes, u32x4 (&v)[NSEG]) {
    const unsigned char* plane_bytes = reinterpret_cast<const unsigned char*>(p.planes);
    const unsigned nlines = R.bw * R.bh;
    const unsigned sWb = (unsigned)p.sW * 4u, sHb = (unsigned)p.sH * 4u;
    // first line of this lane: q = wid*8 + slot  (q < 32 + 8, bw >= 3: float division is exact, see below)
    const unsigned q = (unsigned)wid * 8u + (unsigned)slot;
    const unsigned ry = (unsigned)(((float)q + 0.5f) * (1.0f / (float)R.bw));     // q + 0.5 is never within 1e-3 of a multiple of bw
    unsigned rx = q - ry * R.bw;
    unsigned goff = img_bytes + (unsigned)pl * TT_LINE + (R.y0 - 1u + ry) * sHb + (R.x0 - 1u + rx) * sWb + ch_bytes;
    // +32 lines = a rows + b columns (+1 row -bw columns on wrap)
    const unsigned a = 32u / R.bw, b = 32u - a * R.bw;                              // wave-uniform
    const unsigned inc = a * sHb + b * sWb, inc_wrap = inc + sHb - R.bw * sWb;
#pragma unroll
    for (int j = 0; j < NSEG; ++j) {
        const unsigned q0 = ((unsigned)wid + 4u * j) * 8u;
        if (q0 < nlines) {
            if (q0 + 8u <= nlines || (unsigned)slot < nlines - q0)          // lines past the region would read past its last row
                v[j] = *reinterpret_cast<const u32x4*>(plane_bytes + goff);
            rx += b;
            const bool wrap = rx >= R.bw;
            rx = wrap ? rx - R.bw : rx;
            goff += wrap ? inc_wrap : inc;
        }
    }
}

template <int NSEG>
__device__ __forceinline__ void stage_commit(const Region& R, unsigned char* s_lines, int wid, int lane, const u32x4 (&v)[NSEG]) {
    const unsigned nlines = R.bw * R.bh;
    unsigned char* dst = s_lines + (R.base + (unsigned)wid * 8u) * TT_LINE + (unsigned)lane * 16u;
#pragma unroll
    for (int j = 0; j < NSEG; ++j) {
        const unsigned q0 = ((unsigned)wid + 4u * j) * 8u;
        if (q0 < nlines)
            *reinterpret_cast<u32x4*>(dst + j * 32 * TT_LINE) = v[j];
    }
}


// Staging of the producer / consumer kernel's F waves: the same line -> (wave, segment, lane) mapping, but almost no scalar work
// (the scalar unit is one per CU: the 16 waves of that kernel share it).  Loads are buffer loads bounded by the plane group, so the
// lanes of a region's last, partial segment need no exec mask (they read plane rows below the region, zero beyond the buffer; their
// LDS slots are the padding of the region's 8-line-granular allocation); `ns` = this wave's segment count is the only branch input.
template <int NSEG>
__device__ __forceinline__ int stage_issue_buf(const TileArgs& p, __amdgpu_buffer_rsrc_t rsrc, const Region& R, int pl, unsigned img_bytes,
                                               int wid, int slot, unsigned ch_bytes, u32x4 (&v)[NSEG]) {
    const unsigned nseg8 = (R.bw * R.bh + 7u) >> 3;
    const int ns = (nseg8 > (unsigned)wid) ? (int)((nseg8 - (unsigned)wid + 3u) >> 2) : 0;     // segments wid, wid + 4, ... below nseg8
    const unsigned sWb = (unsigned)p.sW * 4u, sHb = (unsigned)p.sH * 4u;
    const unsigned q = (unsigned)wid * 8u + (unsigned)slot;
    const unsigned ry = (unsigned)(((float)q + 0.5f) * (1.0f / (float)R.bw));
    unsigned rx = q - ry * R.bw;
    unsigned goff = img_bytes + (unsigned)pl * TT_LINE + (R.y0 - 1u + ry) * sHb + (R.x0 - 1u + rx) * sWb + ch_bytes;
    const unsigned a = 32u / R.bw, b = 32u - a * R.bw;
    const unsigned inc = a * sHb + b * sWb, inc_wrap = inc + sHb - R.bw * sWb;
#pragma unroll
    for (int j = 0; j < NSEG; ++j) {
        if (j < ns) {
            v[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)goff, 0, 0));
            rx += b;
            const bool wrap = rx >= R.bw;
            rx = wrap ? rx - R.bw : rx;
            goff += wrap ? inc_wrap : inc;
        }
    }
    return ns;
}

// The same segments by LDS-DMA (`buffer_load_dwordx4 ... lds`: per-lane buffer offsets, the wave's 1 KB lands at M0 + 16 * lane): no
// registers, no ds_write, all three planes of a chunk in flight at once.  Bounded like stage_issue_buf (zeros beyond the plane group).
template <int NSEG, int NW = 4>       // NW: waves that share a region's segments (segment s belongs to wave s % NW)
__device__ __forceinline__ void stage_dma(const TileArgs& p, __amdgpu_buffer_rsrc_t rsrc, const Region& R, int pl, unsigned img_bytes,
                                          int wid, int slot, unsigned ch_bytes, unsigned char* s_lines) {
    const unsigned nseg8 = (R.bw * R.bh + 7u) >> 3;
    const int ns = (nseg8 > (unsigned)wid) ? (int)((nseg8 - (unsigned)wid + (unsigned)(NW - 1)) / (unsigned)NW) : 0;
    const unsigned sWb = (unsigned)p.sW * 4u, sHb = (unsigned)p.sH * 4u;
    const unsigned q = (unsigned)wid * 8u + (unsigned)slot;
    const unsigned ry = (unsigned)(((float)q + 0.5f) * (1.0f / (float)R.bw));
    unsigned rx = q - ry * R.bw;
    unsigned goff = img_bytes + (unsigned)pl * TT_LINE + (R.y0 - 1u + ry) * sHb + (R.x0 - 1u + rx) * sWb + ch_bytes;
    constexpr unsigned STEP = 8u * NW;                                    // lines between two segments of one wave
    const unsigned a = STEP / R.bw, b = STEP - a * R.bw;
    const unsigned inc = a * sHb + b * sWb, inc_wrap = inc + sHb - R.bw * sWb;
    unsigned char* dst = s_lines + (R.base + (unsigned)wid * 8u) * TT_LINE;
#pragma unroll
    for (int j = 0; j < NSEG; ++j) {
        if (j < ns) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + j * (int)STEP * TT_LINE), 16, (int)goff, 0, 0, 0);
            rx += b;
            const bool wrap = rx >= R.bw;
            rx = wrap ? rx - R.bw : rx;
            goff += wrap ? inc_wrap : inc;
        }
    }
}

template <int NSEG>
__device__ __forceinline__ void stage_commit_n(const Region& R, unsigned char* s_lines, int wid, int lane, int ns, const u32x4 (&v)[NSEG]) {
    unsigned char* dst = s_lines + (R.base + (unsigned)wid * 8u) * TT_LINE + (unsigned)lane * 16u;
#pragma unroll
    for (int j = 0; j < NSEG; ++j)
        if (j < ns) *reinterpret_cast<u32x4*>(dst + j * 32 * TT_LINE) = v[j];
}

// Region table of one chunk from the packed bounding boxes of its footprint origins (all wave-uniform): lo0 / hi0 = (x on W |
// y on H << 16), lo1 / hi1 = (y on W | z on H << 16).  Returns the mask of staged planes: all three if they fit `cap` lines, else
// the pair with the smallest footprint that fits (the plane left out is the one with the least reuse), else the smallest single
// plane, else none.
__device__ __forceinline__ unsigned make_regions(unsigned lo0, unsigned lo1, unsigned hi0, unsigned hi1, int W, int H, unsigned cap,
                                                 Region (&R)[3]) {
    const unsigned xw_lo = lo0 & 0xffffu, yh_lo = lo0 >> 16, yw_lo = lo1 & 0xffffu, zh_lo = lo1 >> 16;
    const unsigned xw_hi = hi0 & 0xffffu, yh_hi = hi0 >> 16, yw_hi = hi1 & 0xffffu, zh_hi = hi1 >> 16;
    // footprints are 2 x 2: one more column / row than the span of the origins
    R[0].x0 = xw_lo; R[0].y0 = yh_lo; R[0].bw = (xw_hi - xw_lo + 2u); R[0].bh = yh_hi - yh_lo + 2u;
    R[1].x0 = yw_lo; R[1].y0 = zh_lo; R[1].bw = (yw_hi - yw_lo + 2u); R[1].bh = zh_hi - zh_lo + 2u;
    R[2].x0 = xw_lo; R[2].y0 = zh_lo; R[2].bw = (xw_hi - xw_lo + 2u); R[2].bh = zh_hi - zh_lo + 2u;
    // a region larger than a wave's TT_SEGS segments (or than the budget) counts as "does not fit"
    // ... and so does a region that touches the plane border (virtual column / row 0 or size): staging never clamps
    auto lines_of = [&](const Region& r, unsigned segs) {
        const unsigned l = (r.bw * r.bh + 7u) & ~7u;
        const bool inside = r.x0 >= 1u && r.x0 + r.bw - 2u <= (unsigned)(W - 1) && r.y0 >= 1u && r.y0 + r.bh - 2u <= (unsigned)(H - 1);
        return (inside && l <= segs * 32u) ? l : cap + 1u;
    };
    const unsigned l0 = lines_of(R[0], TT_SEGS_A), l1 = lines_of(R[1], TT_SEGS_B), l2 = lines_of(R[2], TT_SEGS_A);
    unsigned mask;
    if (l0 + l1 + l2 <= cap) mask = 7u;
    else {
        const unsigned s01 = l0 + l1, s02 = l0 + l2, s12 = l1 + l2;
        unsigned best = cap + 1u; mask = 0u;
        if (s01 < best) { best = s01; mask = 3u; }
        if (s02 < best) { best = s02; mask = 5u; }
        if (s12 < best) { best = s12; mask = 6u; }
        if (mask == 0u) {
            if (l0 < best) { best = l0; mask = 1u; }
            if (l1 < best) { best = l1; mask = 2u; }
            if (l2 < best) { best = l2; mask = 4u; }
        }
    }
    unsigned used = 0;
    const unsigned l[3] = {l0, l1, l2};
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) { R[pl].staged = (mask >> pl) & 1u; R[pl].base = used; if (R[pl].staged) used += l[pl]; }
    return mask;
}

// Region table of a chunk as the producer / consumer kernel keeps it in LDS: (x0, y0, bw, bh) x 3 planes, (base x 3, staged mask).
// Lanes 0-3 store one 16-byte row each, picked with selects: written as four whole-vector stores of the (wave-uniform) structs, hipcc
// keeps `R` in scratch memory and reloads it with scratch loads, whose `vmcnt(0)` also waits for the coordinate loads in flight.
__device__ __forceinline__ unsigned pin_s(unsigned x) { asm volatile("" : "+s"(x)); return x; }     // keep a wave-uniform value in an SGPR, opaque to the optimiser
__device__ __forceinline__ void store_regions(u32x4* r, const Region (&R)[3], unsigned mask, int lane) {
    // (without the pins hipcc turns the selects below into an indexed load from a scratch copy of R)
    const unsigned x0[3] = {pin_s(R[0].x0), pin_s(R[1].x0), pin_s(R[2].x0)}, y0[3] = {pin_s(R[0].y0), pin_s(R[1].y0), pin_s(R[2].y0)};
    const unsigned bw[3] = {pin_s(R[0].bw), pin_s(R[1].bw), pin_s(R[2].bw)}, bh[3] = {pin_s(R[0].bh), pin_s(R[1].bh), pin_s(R[2].bh)};
    const unsigned ba[3] = {pin_s(R[0].base), pin_s(R[1].base), pin_s(R[2].base)}, m = pin_s(mask);
    const unsigned a = (lane == 0) ? x0[0] : (lane == 1) ? x0[1] : (lane == 2) ? x0[2] : ba[0];
    const unsigned b = (lane == 0) ? y0[0] : (lane == 1) ? y0[1] : (lane == 2) ? y0[2] : ba[1];
    const unsigned c = (lane == 0) ? bw[0] : (lane == 1) ? bw[1] : (lane == 2) ? bw[2] : ba[2];
    const unsigned d = (lane == 0) ? bh[0] : (lane == 1) ? bh[1] : (lane == 2) ? bh[2] : m;
    if (lane < 4) r[lane] = u32x4{a, b, c, d};
}
// make_regions + store_regions for the producer / consumer kernel with the three planes on lanes 0-2 (round 5): the boxes, line counts and the
// inside / size tests of the three regions are the same arithmetic on different axes, i.e. vector work; only the choice of the staged set and
// the bases need all three line counts (three readlanes, ~25 scalar instructions).  As scalar code the whole table was ~100 dependent SALU
// instructions of a wave whose chain every other wave of the workgroup waits for at the barrier.  Same table as make_regions, bit for bit.
__device__ __forceinline__ void build_region_table(unsigned lo0, unsigned lo1, unsigned hi0, unsigned hi1, int W, int H, unsigned cap, u32x4* table, int lane) {
    const unsigned xw_lo = lo0 & 0xffffu, yh_lo = lo0 >> 16, yw_lo = lo1 & 0xffffu, zh_lo = lo1 >> 16;
    const unsigned xw_hi = hi0 & 0xffffu, yh_hi = hi0 >> 16, yw_hi = hi1 & 0xffffu, zh_hi = hi1 >> 16;
    // plane 0: (x on W, y on H), plane 1: (y on W, z on H), plane 2: (x on W, z on H)
    const unsigned x_lo = (lane == 1) ? yw_lo : xw_lo, x_hi = (lane == 1) ? yw_hi : xw_hi;
    const unsigned y_lo = (lane == 0) ? yh_lo : zh_lo, y_hi = (lane == 0) ? yh_hi : zh_hi;
    const unsigned bw = x_hi - x_lo + 2u, bh = y_hi - y_lo + 2u;
    const unsigned segs = (lane == 1) ? (unsigned)TT_SEGS_B : (unsigned)TT_SEGS_A;
    const unsigned lines = (__umul24(bw, bh) + 7u) & ~7u;
    const bool inside = x_lo >= 1u && x_lo + bw - 2u <= (unsigned)(W - 1) && y_lo >= 1u && y_lo + bh - 2u <= (unsigned)(H - 1);
    const unsigned l = (inside && lines <= segs * 32u) ? lines : cap + 1u;
    const unsigned l0 = (unsigned)__builtin_amdgcn_readlane((int)l, 0), l1 = (unsigned)__builtin_amdgcn_readlane((int)l, 1), l2 = (unsigned)__builtin_amdgcn_readlane((int)l, 2);
    unsigned mask;
    if (l0 + l1 + l2 <= cap) mask = 7u;
    else {
        const unsigned s01 = l0 + l1, s02 = l0 + l2, s12 = l1 + l2;
        unsigned best = cap + 1u; mask = 0u;
        if (s01 < best) { best = s01; mask = 3u; }
        if (s02 < best) { best = s02; mask = 5u; }
        if (s12 < best) { best = s12; mask = 6u; }
        if (mask == 0u) {
            if (l0 < best) { best = l0; mask = 1u; }
            if (l1 < best) { best = l1; mask = 2u; }
            if (l2 < best) { best = l2; mask = 4u; }
        }
    }
    const unsigned b1 = (mask & 1u) ? l0 : 0u, b2 = b1 + ((mask & 2u) ? l1 : 0u);
    const u32x4 row = (lane < 3) ? u32x4{x_lo, y_lo, bw, bh} : u32x4{0u, b1, b2, mask};
    if (lane < 4) table[lane] = row;
}

__device__ __forceinline__ unsigned load_regions(const u32x4* r, Region (&R)[3]) {
    const u32x4 bases = r[3];
    const unsigned mask = uni(bases[3]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const u32x4 v = r[pl];
        R[pl].x0 = uni(v[0]); R[pl].y0 = uni(v[1]); R[pl].bw = uni(v[2]); R[pl].bh = uni(v[3]);
        R[pl].base = uni(bases[pl]); R[pl].staged = (mask >> pl) & 1u;
    }
    return mask;
}

// Tap table entry of the sample this lane set up: the masked weights per plane + the byte offset of its nw line per plane (an
// LDS offset for staged planes, else a plane offset with "x / y neighbour is another texel" in bits 0 / 1).
__device__ __forceinline__ void write_tap_entry(const TileArgs& p, const SampleTaps& t, const Region (&R)[3], unsigned img_bytes,
                                                u32x4 (*entry)) {
    unsigned o[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const PlaneTap& q = t.pl[pl];
        if (R[pl].staged) {
            o[pl] = (R[pl].base + __umul24(q.vy - R[pl].y0, R[pl].bw) + (q.vx - R[pl].x0)) * TT_LINE;      // < 2^16 each: full-rate 24-bit multiply
        } else {
            // clamped texel (virtual index - 1 clamped into the plane) + "neighbour is another texel" flags
            const unsigned x0c = max(q.vx, 1u) - 1u, x1c = min(q.vx, (unsigned)p.W - 1u);
            const unsigned y0c = max(q.vy, 1u) - 1u, y1c = min(q.vy, (unsigned)p.H - 1u);
            o[pl] = (img_bytes + (unsigned)pl * TT_LINE + (y0c * (unsigned)p.sH + x0c * (unsigned)p.sW) * 4u) |
                    (x1c != x0c ? 1u : 0u) | (y1c != y0c ? 2u : 0u);
        }
        entry[pl] = u32x4{__float_as_uint(q.w00), __float_as_uint(q.w01), __float_as_uint(q.w10), __float_as_uint(q.w11)};
    }
    entry[3] = u32x4{o[0], o[1], o[2], 0u};
}

__global__ void __launch_bounds__(256, 2)
triplane_sample_tile_kernel(const TileArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char s_lines[TT_CAP * TT_LINE];
    __shared__ __attribute__((aligned(16))) u32x4 s_tap[256][4];
    __shared__ unsigned s_bb[4][4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = lane >> 3, cl = lane & 7;
    const unsigned ch_bytes = (unsigned)cl * 16u;

    const unsigned blk = (unsigned)xcd_remap(blockIdx.x, gridDim.x);
    const unsigned seg = blk % (unsigned)p.segs;
    const unsigned tile_lin = blk / (unsigned)p.segs;
    const unsigned img = tile_lin / (unsigned)p.tiles_per_image;
    const unsigned tile = tile_lin - img * (unsigned)p.tiles_per_image;
    const unsigned ty = tile / (unsigned)p.tiles_x, tx = tile - ty * (unsigned)p.tiles_x;
    const unsigned img_bytes = img * p.sN_bytes;
    const unsigned ray00 = ty * TT_EDGE * (unsigned)p.rays_w + tx * TT_EDGE;      // first ray of the tile
    const unsigned step_begin = seg * (unsigned)p.chunks_per_seg * TT_DS;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.planes, 0, (int)p.group_bytes, 0x00020000);

    // phase-A sample of this lane: (ray rl of the tile, depth ds of the chunk)
    const unsigned rl = (unsigned)tid >> 2, ds = (unsigned)tid & 3u;
    const unsigned ray_a = ray00 + (rl >> 3) * (unsigned)p.rays_w + (rl & 7u);
    const unsigned row_a = (img * (unsigned)p.rays_per_image + ray_a) * (unsigned)p.steps;
    const unsigned last_step = (unsigned)p.steps - 1u;

    // Software pipeline over chunks: the taps of chunk k+1 are computed (and the coordinates of chunk k+2 requested) while
    // chunk k is blended, so the top of the loop depends on no outstanding vector-memory operation — in particular not on
    // the previous chunk's output stores, which share `vmcnt` with loads on gfx9.
    const int W = p.W, H = p.H;
    auto coord_ptr = [&](unsigned step) { return p.coords + (size_t)(row_a + min(step + ds, last_step)) * 3; };
    SampleTaps t;
    float cx, cy, cz;
    {
        const float* cp = coord_ptr(step_begin);
        t = sample_taps(cp[0], cp[1], cp[2], W, H);
        const float* np_ = coord_ptr(step_begin + TT_DS);
        cx = np_[0]; cy = np_[1]; cz = np_[2];
    }
    for (int ch = 0; ch < p.chunks_per_seg; ++ch) {
        IDE3D_TS(0)
        const unsigned step0 = step_begin + (unsigned)ch * TT_DS;
        // ---- A: bounding boxes of this chunk's footprints (per axis: x on W, y on H, y on W, z on H) ----------------
        {
            const unsigned lo0 = wave_reduce_pk<false>(t.ax[0]), lo1 = wave_reduce_pk<false>(t.ax[1]);
            const unsigned hi0 = wave_reduce_pk<true>(t.ax[0]), hi1 = wave_reduce_pk<true>(t.ax[1]);
            if (lane == 0) { s_bb[wid][0] = lo0; s_bb[wid][1] = lo1; s_bb[wid][2] = hi0; s_bb[wid][3] = hi1; }
        }
        // LDS-only barrier (no vmcnt drain): s_bb visible, and every wave has finished blending the previous chunk, so
        // s_lines may be overwritten.
        IDE3D_TS(1)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        IDE3D_TS(2)
        // ---- region table (wave-uniform) ----------------------------------------------------------------------
        Region R[3];
        const unsigned mask = make_regions(
            uni(pk_min(pk_min(s_bb[0][0], s_bb[1][0]), pk_min(s_bb[2][0], s_bb[3][0]))),
            uni(pk_min(pk_min(s_bb[0][1], s_bb[1][1]), pk_min(s_bb[2][1], s_bb[3][1]))),
            uni(pk_max(pk_max(s_bb[0][2], s_bb[1][2]), pk_max(s_bb[2][2], s_bb[3][2]))),
            uni(pk_max(pk_max(s_bb[0][3], s_bb[1][3]), pk_max(s_bb[2][3], s_bb[3][3]))), W, H, (unsigned)TT_CAP, R);
        // ---- B: fetch the bounding boxes: planes 0 and 1 now, plane 2 once plane 0 is in LDS (two register sets) ----
        u32x4 sva[TT_SEGS_A], svb[TT_SEGS_B];
        if (R[0].staged) stage_issue(p, R[0], 0, img_bytes, wid, slot, ch_bytes, sva);
        if (R[1].staged) stage_issue(p, R[1], 1, img_bytes, wid, slot, ch_bytes, svb);
        IDE3D_TS(3)
        // ---- tap table -------------------------------------------------------------------------------------------
        const unsigned pitch[3] = {R[0].bw * (unsigned)TT_LINE, R[1].bw * (unsigned)TT_LINE, R[2].bw * (unsigned)TT_LINE};
        write_tap_entry(p, t, R, img_bytes, s_tap[tid]);
        if (R[0].staged) stage_commit(R[0], s_lines, wid, lane, sva);
        if (R[2].staged) stage_issue(p, R[2], 2, img_bytes, wid, slot, ch_bytes, sva);
        if (R[1].staged) stage_commit(R[1], s_lines, wid, lane, svb);
        if (R[2].staged) stage_commit(R[2], s_lines, wid, lane, sva);
        IDE3D_TS(4)
        __syncthreads();      // regions and taps visible; next coordinates arrived (the barrier's release drains vmcnt)
        IDE3D_TS(5)
        // ---- C: blend --------------------------------------------------------------------------------------------
        switch (mask) {
#define IDE3D_BLEND(M) case M: blend_chunk<M>(p, s_lines, s_tap, rsrc, pitch, wid, slot, cl, ray00, img, step0); break;
        IDE3D_BLEND(7) IDE3D_BLEND(6) IDE3D_BLEND(5) IDE3D_BLEND(3) IDE3D_BLEND(4) IDE3D_BLEND(2) IDE3D_BLEND(1)
        default: blend_chunk<0>(p, s_lines, s_tap, rsrc, pitch, wid, slot, cl, ray00, img, step0); break;
#undef IDE3D_BLEND
        }
        IDE3D_TS(6)
        // next chunk: taps now (its coordinates arrived before the barrier above), coordinates of the chunk after it requested
        t = sample_taps(cx, cy, cz, W, H);
        {
            const float* np_ = coord_ptr(step0 + 2 * TT_DS);
            cx = np_[0]; cy = np_[1]; cz = np_[2];
        }
    }
}


// ---- producer / consumer form (round 3) ------------------------------------------------------------------------------
// The 4-wave kernel above runs its phases back to back: two workgroups per CU are supposed to overlap one group's fetch with the
// other's blend, but they fall into step (both blend, then both fetch), so the LDS pipe is ~46 % busy and nothing else is saturated
// (DESIGN.md section 5.2).  Here ONE workgroup owns the CU and every phase has its own waves, one of each kind per SIMD:
//   T  waves 0-3   "taps":    coordinates -> taps, bounding boxes (exchanged among the four T waves through LDS with a monotonic
//                             arrival counter — no workgroup barrier, the other roles are never involved), region table, tap table
//   F  waves 4-7   "fetch":   region loads (global_load_dwordx4 -> registers) and the LDS fill, nothing else
//   B  waves 8..   "blend":   phase C, 4 or 8 waves
// working on chunks k + 2, k + 1 and k of the same ray tile in iteration k; lines and tap tables are double-buffered and one
// `s_barrier` per iteration hands everything over.  Same arithmetic, same tap table, same blend code as above: bit-equal results.
// (First attempt, measured: stager waves doing T + F in sequence beside four blenders = 7.0k cycles per chunk for the stager against
// 5.3k for the blender, 80.5 us — the same as the 4-wave kernel; the split below takes the fetch off the tap waves' critical path.)
// Alternatives that were built and measured (profiles/round5/gather_experiments.txt); the constants select what is kept, the compiler drops the rest:
constexpr bool IDE3D_PC_RWAVE = false;        // true: wave 7 is a dedicated region builder R (F = waves 4-6) — measured 1-2 us SLOWER (T wave 0 builds the table beside its taps): R alone needs a whole iteration for the table chain and is last at the barrier instead
constexpr bool IDE3D_PC_VECTABLE = true;      // the region table's per-plane arithmetic on lanes 0-2 (build_region_table); false: all scalar (round 3)
constexpr bool IDE3D_PC_DMA = true;           // the F waves fill the line buffers by LDS-DMA (stage_dma) instead of loads + ds_write
constexpr int IDE3D_PC_PRIO_T = 0, IDE3D_PC_PRIO_T0 = 0, IDE3D_PC_PRIO_F = 0, IDE3D_PC_PRIO_B = 0;      // s_setprio per role: no effect in any combination

#ifdef IDE3D_TT_TRACE
#ifndef IDE3D_PC_TRACE_BLOCK
#define IDE3D_PC_TRACE_BLOCK 100
#endif
__device__ unsigned long long g_pc_dbg[3][32][8];
__device__ unsigned long long g_pc_wg[1024][4];          // per workgroup (B wave 0): shader cycles entry -> end, 100 MHz clock at entry / end, staged-plane masks seen
#define IDE3D_PCT(role, k) if (blockIdx.x == IDE3D_PC_TRACE_BLOCK && lane == 0 && ridx == 0 && it >= 0 && it < 32) g_pc_dbg[role][it][(k)] = __builtin_readcyclecounter();
#else
#define IDE3D_PCT(role, k)
#endif

// = axis_tap().v: the virtual index of the low tap of one axis (the index half of the tap arithmetic)
__device__ __forceinline__ unsigned axis_index(float c, int size) {
    const float fu = floorf(unnormalize(c, size));
    return (unsigned)(min(max((int)fminf(fmaxf(fu, -2.0f), (float)size + 1.0f), -1), size - 1) + 1);
}

template <int NB, int FR>      // NB: blender waves (4 or 8); FR: fetch rounds (1 = all three planes in flight at once, 2 = planes 0 + 1, then 2)
__global__ void __launch_bounds__(64 * (8 + NB), (8 + NB) / 4)
triplane_sample_tile_pc_kernel(const TileArgs p) {
    // one LDS object: [2][PC_CAP lines] [2][256 tap entries] [2][16] region tables, [2][4] blend descriptors
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PC_CAP * TT_LINE + 2 * 256 * 64 + 512];
    unsigned char* const s_lines0 = smem;
    u32x4 (*const s_tap0)[4] = reinterpret_cast<u32x4 (*)[4]>(smem + 2 * PC_CAP * TT_LINE);
    unsigned* const s_misc = reinterpret_cast<unsigned*>(smem + 2 * PC_CAP * TT_LINE + 2 * 256 * 64);
    u32x4 (*const s_reg)[4] = reinterpret_cast<u32x4 (*)[4]>(s_misc + 32);                  // [2][4]: (x0, y0, bw, bh) x 3, (base x 3, mask)
    u32x4* const s_meta = reinterpret_cast<u32x4*>(s_misc + 64);                            // [2]: mask, bw0, bw1, bw2

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ridx = (wid < 8) ? (wid & 3) : wid - 8;                     // wave index inside its role
    const int slot = lane >> 3, cl = lane & 7;
    const unsigned ch_bytes = (unsigned)cl * 16u;

    const unsigned blk = (unsigned)xcd_remap(blockIdx.x, gridDim.x);
    const unsigned seg = blk % (unsigned)p.segs;
    const unsigned tile_lin = blk / (unsigned)p.segs;
    const unsigned img = tile_lin / (unsigned)p.tiles_per_image;
    const unsigned tile = tile_lin - img * (unsigned)p.tiles_per_image;
    const unsigned ty = tile / (unsigned)p.tiles_x, tx = tile - ty * (unsigned)p.tiles_x;
    const unsigned img_bytes = img * p.sN_bytes;
    const unsigned ray00 = ty * TT_EDGE * (unsigned)p.rays_w + tx * TT_EDGE;
    const unsigned step_begin = seg * (unsigned)p.chunks_per_seg * TT_DS;
    const int nch = p.chunks_per_seg;
    const int W = p.W, H = p.H;

#ifdef IDE3D_TT_TRACE
    const unsigned long long wg_c0 = __builtin_readcyclecounter(), wg_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long wg_masks = 0;
#endif
    __syncthreads();

    if (wid < 4) {
        // ---------------------------------------------------------------- T: taps, bounding boxes, region + tap tables ----
        if (IDE3D_PC_PRIO_T) __builtin_amdgcn_s_setprio(IDE3D_PC_PRIO_T);
        if (IDE3D_PC_PRIO_T0 && ridx == 0) __builtin_amdgcn_s_setprio(IDE3D_PC_PRIO_T0);          // the wave that builds the region table
        // phase-A sample of this lane: (ray rl of the tile, depth ds of the chunk); t_id = index among the 256 samples of a chunk
        const unsigned t_id = (unsigned)ridx * 64u + (unsigned)lane;
        const unsigned rl = t_id >> 2, ds = t_id & 3u;
        const unsigned ray_a = ray00 + (rl >> 3) * (unsigned)p.rays_w + (rl & 7u);
        const unsigned row_a = (img * (unsigned)p.rays_per_image + ray_a) * (unsigned)p.steps;
        const unsigned last_step = (unsigned)p.steps - 1u;
        auto coord_ptr = [&](unsigned step) { return p.coords + (size_t)(row_a + min(step + ds, last_step)) * 3; };
        // Region table of a chunk: wave 0 alone builds it, from the footprint origins of ALL 256 samples — its own 64 from the taps it
        // sets up anyway, the other 192 from the index half of the tap arithmetic on coordinates it loads itself (3 more samples per
        // lane).  No exchange between the T waves: an LDS round trip takes hundreds of cycles while eight blending waves keep the LDS
        // queue full, and the first version (boxes posted to LDS, arrival counter, poll) spent 3 - 4k cycles per chunk there.  The
        // table is scalar work (the scalar unit is one per CU and 16 waves share it); F reads it one iteration later, T two later.
        auto build_regions = [&](const SampleTaps& t, const float (&oc)[3][3], u32x4* table) {
            unsigned lo0 = t.ax[0], lo1 = t.ax[1], hi0 = t.ax[0], hi1 = t.ax[1];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const unsigned a0 = axis_index(oc[k][0], W) | (axis_index(oc[k][1], H) << 16);
                const unsigned a1 = axis_index(oc[k][1], W) | (axis_index(oc[k][2], H) << 16);
                lo0 = pk_min(lo0, a0); hi0 = pk_max(hi0, a0); lo1 = pk_min(lo1, a1); hi1 = pk_max(hi1, a1);
            }
            wave_reduce_pk4(lo0, lo1, hi0, hi1);
            if (IDE3D_PC_VECTABLE) build_region_table(lo0, lo1, hi0, hi1, W, H, (unsigned)PC_CAP, table, lane);
            else {
                Region R[3];
                const unsigned mask = make_regions(lo0, lo1, hi0, hi1, W, H, (unsigned)PC_CAP, R);
                store_regions(table, R, mask, lane);
            }
        };
        // coordinates of the other three waves' samples (wave 0 only): sample t_id + 64 k of the chunk
        auto other_coord_ptr = [&](int k, unsigned step) {
            const unsigned t2 = t_id + 64u * (unsigned)k, rl2 = t2 >> 2, ds2 = t2 & 3u;
            const unsigned ray2 = ray00 + (rl2 >> 3) * (unsigned)p.rays_w + (rl2 & 7u);
            return p.coords + (size_t)((img * (unsigned)p.rays_per_image + ray2) * (unsigned)p.steps + min(step + ds2, last_step)) * 3;
        };
        float oc[3][3] = {};
        if (!IDE3D_PC_RWAVE && ridx == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { const float* q = other_coord_ptr(k + 1, step_begin); oc[k][0] = q[0]; oc[k][1] = q[1]; oc[k][2] = q[2]; }
        }
        float cx, cy, cz;
        {
            const float* cp = coord_ptr(step_begin);
            cx = cp[0]; cy = cp[1]; cz = cp[2];
        }
        SampleTaps t_hold;
        for (int it = -2; it < nch; ++it) {
            IDE3D_PCT(0, 0)
            // tap table + blend descriptor of chunk it + 1 (taps set up one iteration ago — its buffer was still being blended from
            // then; region table built by wave 0 in that iteration)
            if (it + 1 >= 0 && it + 1 < nch) {
                const unsigned buf = (unsigned)(it + 1) & 1u;
                Region R[3];
                const unsigned mask = load_regions(s_reg[buf], R);
                write_tap_entry(p, t_hold, R, img_bytes, s_tap0[buf * 256u + t_id]);
                if (t_id == 0) s_meta[buf] = u32x4{mask, R[0].bw, R[1].bw, R[2].bw};
            }
            IDE3D_PCT(0, 1)
            // chunk it + 2: taps (its coordinates were requested one iteration ago), coordinates of the chunk after it, boxes, regions
            if (it + 2 < nch) {
                t_hold = sample_taps(cx, cy, cz, W, H);
                float oc_now[3][3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { oc_now[k][0] = oc[k][0]; oc_now[k][1] = oc[k][1]; oc_now[k][2] = oc[k][2]; }
                {
                    const float* np_ = coord_ptr(step_begin + (unsigned)(it + 3) * TT_DS);
                    cx = np_[0]; cy = np_[1]; cz = np_[2];
                    if (!IDE3D_PC_RWAVE && ridx == 0) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) { const float* q = other_coord_ptr(k + 1, step_begin + (unsigned)(it + 3) * TT_DS); oc[k][0] = q[0]; oc[k][1] = q[1]; oc[k][2] = q[2]; }
                    }
                }
                IDE3D_PCT(0, 2)
                if (!IDE3D_PC_RWAVE && ridx == 0) build_regions(t_hold, oc_now, s_reg[(unsigned)it & 1u]);
            }
            IDE3D_PCT(0, 3)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            IDE3D_PCT(0, 4)
        }
    } else if (IDE3D_PC_RWAVE && wid == 7) {
        // ---------------------------------------------------------------- R: region table (round 5) ----------------------------
        // Wave 7 builds the region table of chunk it + 2 and does nothing else: the footprint origins of all 256 samples (4 per lane,
        // from coordinates it loads itself one iteration ahead), the wave-wide reduction, the scalar table code — 2.5k cycles per chunk
        // that used to sit on T wave 0 BEHIND its own taps and tap-table entries, which made that wave the last at the barrier in most
        // iterations (gpurun_out -> profiles/round5/gather_experiments.txt).  The fill needs three waves, not four, since it is LDS-DMA.
        const unsigned last_step = (unsigned)p.steps - 1u;
        auto sample_ptr = [&](int k, unsigned step) {          // sample lane + 64 k of a chunk = (tile ray, depth)
            const unsigned t2 = (unsigned)lane + 64u * (unsigned)k, rl2 = t2 >> 2, ds2 = t2 & 3u;
            const unsigned ray2 = ray00 + (rl2 >> 3) * (unsigned)p.rays_w + (rl2 & 7u);
            return p.coords + (size_t)((img * (unsigned)p.rays_per_image + ray2) * (unsigned)p.steps + min(step + ds2, last_step)) * 3;
        };
        float c[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float* q = sample_ptr(k, step_begin); c[k][0] = q[0]; c[k][1] = q[1]; c[k][2] = q[2]; }
        for (int it = -2; it < nch; ++it) {
            if (it + 2 < nch) {
                float cn[4][3];
#pragma unroll
                for (int k = 0; k < 4; ++k) { cn[k][0] = c[k][0]; cn[k][1] = c[k][1]; cn[k][2] = c[k][2]; }
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float* q = sample_ptr(k, step_begin + (unsigned)(it + 3) * TT_DS); c[k][0] = q[0]; c[k][1] = q[1]; c[k][2] = q[2]; }
                unsigned lo0 = 0xffffffffu, lo1 = 0xffffffffu, hi0 = 0u, hi1 = 0u;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned a0 = axis_index(cn[k][0], W) | (axis_index(cn[k][1], H) << 16);
                    const unsigned a1 = axis_index(cn[k][1], W) | (axis_index(cn[k][2], H) << 16);
                    lo0 = pk_min(lo0, a0); hi0 = pk_max(hi0, a0); lo1 = pk_min(lo1, a1); hi1 = pk_max(hi1, a1);
                }
                wave_reduce_pk4(lo0, lo1, hi0, hi1);
                if (IDE3D_PC_VECTABLE) build_region_table(lo0, lo1, hi0, hi1, W, H, (unsigned)PC_CAP, s_reg[(unsigned)it & 1u], lane);
                else {
                    Region R[3];
                    const unsigned mask = make_regions(lo0, lo1, hi0, hi1, W, H, (unsigned)PC_CAP, R);
                    store_regions(s_reg[(unsigned)it & 1u], R, mask, lane);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else if (wid < 8) {
        // ---------------------------------------------------------------- F: region loads + LDS fill ------------------------
        if (IDE3D_PC_PRIO_F) __builtin_amdgcn_s_setprio(IDE3D_PC_PRIO_F);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.planes, 0, (int)p.group_bytes, 0x00020000);
        for (int it = -2; it < nch; ++it) {
            IDE3D_PCT(1, 0)
            if (it + 1 >= 0 && it + 1 < nch) {
                const unsigned buf = (unsigned)(it + 1) & 1u;
                unsigned char* const s_lines = s_lines0 + buf * (PC_CAP * TT_LINE);
                Region R[3];
                load_regions(s_reg[buf], R);
                // (Tried: the F waves touching one dword per line of the NOT staged planes, to pull them into L1 / L2 ahead of the blending
                // waves' buffer loads — 64 separate lines per load instruction made F the slowest role: 73.5 vs 70.3 us.)
                int n0 = 0, n1 = 0, n2 = 0;
                if (IDE3D_PC_DMA && IDE3D_PC_RWAVE) {          // three fetching waves (4-6)
                    if (R[0].staged) stage_dma<(TT_SEGS_A * 4 + 2) / 3, 3>(p, rsrc, R[0], 0, img_bytes, ridx, slot, ch_bytes, s_lines);
                    if (R[1].staged) stage_dma<(TT_SEGS_B * 4 + 2) / 3, 3>(p, rsrc, R[1], 1, img_bytes, ridx, slot, ch_bytes, s_lines);
                    if (R[2].staged) stage_dma<(TT_SEGS_A * 4 + 2) / 3, 3>(p, rsrc, R[2], 2, img_bytes, ridx, slot, ch_bytes, s_lines);
                    IDE3D_PCT(1, 1)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else if (IDE3D_PC_DMA) {
                    if (R[0].staged) stage_dma<TT_SEGS_A>(p, rsrc, R[0], 0, img_bytes, ridx, slot, ch_bytes, s_lines);
                    if (R[1].staged) stage_dma<TT_SEGS_B>(p, rsrc, R[1], 1, img_bytes, ridx, slot, ch_bytes, s_lines);
                    if (R[2].staged) stage_dma<TT_SEGS_A>(p, rsrc, R[2], 2, img_bytes, ridx, slot, ch_bytes, s_lines);
                    IDE3D_PCT(1, 1)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else if (FR == 1) {
                    u32x4 sva[TT_SEGS_A], svb[TT_SEGS_B], svc[TT_SEGS_A];
                    if (R[0].staged) n0 = stage_issue_buf(p, rsrc, R[0], 0, img_bytes, ridx, slot, ch_bytes, sva);
                    if (R[1].staged) n1 = stage_issue_buf(p, rsrc, R[1], 1, img_bytes, ridx, slot, ch_bytes, svb);
                    if (R[2].staged) n2 = stage_issue_buf(p, rsrc, R[2], 2, img_bytes, ridx, slot, ch_bytes, svc);
                    IDE3D_PCT(1, 1)
                    stage_commit_n(R[0], s_lines, ridx, lane, n0, sva);
                    stage_commit_n(R[1], s_lines, ridx, lane, n1, svb);
                    stage_commit_n(R[2], s_lines, ridx, lane, n2, svc);
                } else {
                    u32x4 sva[TT_SEGS_A], svb[TT_SEGS_B];
                    if (R[0].staged) n0 = stage_issue_buf(p, rsrc, R[0], 0, img_bytes, ridx, slot, ch_bytes, sva);
                    if (R[1].staged) n1 = stage_issue_buf(p, rsrc, R[1], 1, img_bytes, ridx, slot, ch_bytes, svb);
                    IDE3D_PCT(1, 1)
                    stage_commit_n(R[0], s_lines, ridx, lane, n0, sva);
                    if (R[2].staged) n2 = stage_issue_buf(p, rsrc, R[2], 2, img_bytes, ridx, slot, ch_bytes, sva);
                    stage_commit_n(R[1], s_lines, ridx, lane, n1, svb);
                    stage_commit_n(R[2], s_lines, ridx, lane, n2, sva);
                }
            }
            IDE3D_PCT(1, 2)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            IDE3D_PCT(1, 3)
        }
    } else {
        // ---------------------------------------------------------------- B: blend --------------------------------------------
        if (IDE3D_PC_PRIO_B) __builtin_amdgcn_s_setprio(IDE3D_PC_PRIO_B);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.planes, 0, (int)p.group_bytes, 0x00020000);
        for (int it = -2; it < nch; ++it) {
            IDE3D_PCT(2, 0)
            if (it >= 0) {
                const unsigned buf = (unsigned)it & 1u;
                const unsigned step0 = step_begin + (unsigned)it * TT_DS;
                const u32x4 m = s_meta[buf];
                const unsigned mask = uni(m[0]);
#ifdef IDE3D_TT_TRACE
                wg_masks += (mask == 7u) ? 1ull : (1ull << 32);          // low word: chunks with all planes staged; high word: the others
#endif
                const unsigned pitch[3] = {uni(m[1]) * (unsigned)TT_LINE, uni(m[2]) * (unsigned)TT_LINE, uni(m[3]) * (unsigned)TT_LINE};
                const unsigned char* s_lines = s_lines0 + buf * (PC_CAP * TT_LINE);
                const u32x4 (*s_tap)[4] = s_tap0 + buf * 256u;
                switch (mask) {
#define IDE3D_BLEND(M) case M: blend_chunk<M, 32 / NB>(p, s_lines, s_tap, rsrc, pitch, ridx, slot, cl, ray00, img, step0); break;
                IDE3D_BLEND(7) IDE3D_BLEND(6) IDE3D_BLEND(5) IDE3D_BLEND(3) IDE3D_BLEND(4) IDE3D_BLEND(2) IDE3D_BLEND(1)
                default: blend_chunk<0, 32 / NB>(p, s_lines, s_tap, rsrc, pitch, ridx, slot, cl, ray00, img, step0); break;
#undef IDE3D_BLEND
                }
            }
            IDE3D_PCT(2, 1)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            IDE3D_PCT(2, 2)
        }
#ifdef IDE3D_TT_TRACE
        if (ridx == 0 && lane == 0 && blockIdx.x < 1024) {
            g_pc_wg[blockIdx.x][0] = __builtin_readcyclecounter() - wg_c0;
            g_pc_wg[blockIdx.x][1] = wg_r0; g_pc_wg[blockIdx.x][2] = __builtin_amdgcn_s_memrealtime();
            g_pc_wg[blockIdx.x][3] = wg_masks;
        }
#endif
    }
}

}  // namespace

// Returns false when the hint / layout does not fit this kernel (the caller then uses the flat kernel).
bool launch_triplane_tile(const float* planes, const int64_t* s, int n, int C, int H, int W,
                          const float* coords, int64_t m, float* out,
                          int rays_h, int rays_w, int steps, hipStream_t st) {
    if (C != TT_C || rays_h <= 0 || rays_w <= 0 || steps <= 0) return false;
    if ((int64_t)rays_h * rays_w * steps != m) return false;
    if (rays_h % TT_EDGE || rays_w % TT_EDGE || steps % TT_DS) return false;
    if (H > 0x7fff || W > 0x7fff) return false;
    if (s[1] != 1 || s[3] < TT_C * 3 || s[2] <= 0) return false;          // three 128-byte lines per pixel, channels innermost
    const int64_t sN_bytes = s[0] * 4;
    if (sN_bytes <= 0 || sN_bytes >= (1LL << 31) || (int64_t)H * s[2] * 4 > sN_bytes) return false;
    int group = (int)(((1LL << 31) - 1) / sN_bytes);
    while (group >= 1 && (int64_t)group * m >= (1LL << 31)) group >>= 1;
    if (group < 1) return false;
    const int chunks = steps / TT_DS;
    const int tiles_per_image = (rays_h / TT_EDGE) * (rays_w / TT_EDGE);
    const int env_segs = knobs().gather_segs;
    // IDE3D_GATHER_PC=0: the 4-wave kernel with two workgroups per CU (rounds 1-2)
    const int pc_form = knobs().gather_pc;      // 4 / 8: blending waves
    const bool use_pc = pc_form == 4 || pc_form == 8;
    for (int n0 = 0; n0 < n; n0 += group) {
        const int cnt = (n - n0 < group) ? n - n0 : group;
        // depth segments: enough workgroups for >= 2 per CU in flight on every CU, as few as possible otherwise
        int segs = 1;
        while (segs < chunks && ((int64_t)cnt * tiles_per_image * segs < 2 * kNumCU || chunks % segs)) ++segs;
        if (env_segs > 0 && chunks % env_segs == 0) segs = env_segs;
        TileArgs a;
        a.planes = planes + (int64_t)n0 * s[0];
        a.coords = coords + (int64_t)n0 * m * 3;
        a.out = out + (int64_t)n0 * m * C;
        a.sN_bytes = (unsigned)sN_bytes; a.group_bytes = (unsigned)(cnt * sN_bytes);
        a.sH = (int)s[2]; a.sW = (int)s[3]; a.H = H; a.W = W;
        a.rays_w = rays_w; a.rays_per_image = rays_h * rays_w; a.steps = steps;
        a.tiles_x = rays_w / TT_EDGE; a.tiles_per_image = tiles_per_image; a.segs = segs; a.chunks_per_seg = chunks / segs;
        if (use_pc) {
            // producer / consumer form: one 8-wave workgroup per CU; depth segments only until every CU has one
            int ps = 1;
            while (ps < chunks && ((int64_t)cnt * tiles_per_image * ps < kNumCU || chunks % ps)) ++ps;
            if (env_segs > 0 && chunks % env_segs == 0) ps = env_segs;
            a.segs = ps; a.chunks_per_seg = chunks / ps;
            const dim3 grid((unsigned)(cnt * tiles_per_image * ps));
            if (pc_form == 8) hipLaunchKernelGGL((triplane_sample_tile_pc_kernel<8, 2>), grid, dim3(1024), 0, st, a);
            else hipLaunchKernelGGL((triplane_sample_tile_pc_kernel<4, 1>), grid, dim3(768), 0, st, a);
        } else
        hipLaunchKernelGGL(triplane_sample_tile_kernel, dim3((unsigned)(cnt * tiles_per_image * segs)), dim3(256), 0, st, a);
    }
    return true;
}

}  // namespace ide3d

#ifdef IDE3D_TT_TRACE
extern "C" int ide3d_debug_tt(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ide3d::g_tt_dbg), sizeof(unsigned long long) * 256);
}
extern "C" int ide3d_debug_tt_wg(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ide3d::g_pc_wg), sizeof(unsigned long long) * 1024 * 4);
}
extern "C" int ide3d_debug_tt_pc(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ide3d::g_pc_dbg), sizeof(unsigned long long) * 3 * 32 * 8);
}
#endif

namespace ide3d {
const char* triplane_tile_build_flags() {
    return ""
#ifdef IDE3D_TT_TRACE
        "IDE3D_TT_TRACE "
#endif
        ;
}
}  // namespace ide3d
