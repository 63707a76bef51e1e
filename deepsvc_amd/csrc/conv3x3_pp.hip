// conv3x3_pp.hip -- "ping-pong" f16 3x3 stride-1 convolution, 64 input channels -> 32/64 output channels, f16 NHWC in
// and out, activation in {none, ReLU, LeakyReLU} on either side, optional f16 residual(s): the ResBlock1 / in_conv
// family that carries ~70 % of a 1080p P-frame's MACs (modules.py:200-204,307-311,322-340,382-418; census: 70
// launches per frame at 1088x1920, tools/ws_census.py).
//
// Why a second kernel next to conv3x3_ws.hip: measured by skipping phases, that kernel's
// time is the SUM of its MFMA phase (122 us), patch staging (60 us) and epilogue (67 us, 210 us with a residual):
// its 8 waves move through the phases in lock step, so the matrix pipes idle whenever the waves stage or store.
// Here a workgroup is TWO groups of 4 waves (one wave of each group on every SIMD), each group walking its own
// 8x32-pixel tiles with its own halo-patch buffer, and group 1 runs ONE barrier phase behind group 0:
//
//     phase        1        2        3        4        1 ...
//     group 0      M0       S1       M1       E+S0     M0          M<c> = 72 MFMAs on input chunk c (32 channels)
//     group 1      E+S0     M0       S1       M1       E+S0        S<c> = write the prefetched chunk-c patch to LDS
//                                                                  E    = fused epilogue of the finished tile
//
// so in every phase exactly one wave per SIMD feeds the matrix pipe while its partner does the vector / memory work.
// The s_barrier is workgroup-wide; both groups execute the same number of barriers, group 1 simply starts with an
// extra one and group 0 ends with one.
//   * weights: all 2 chunks x 9 taps x Cout rows resident in LDS (80-B pitch), staged once per workgroup;
//   * patch: 10x34 pixels x one 64-B chunk per group (80-B pitch: conflict-free ds_read_b128); the next chunk is
//     prefetched into registers with bounds-checked buffer loads at the START of an M phase and written in the
//     following S phase, so HBM/L2 latency rides under 72 MFMAs;
//   * residual vectors of the tile are prefetched the same way at the start of M1;
//   * epilogue WITHOUT LDS (a first version went through LDS slabs: its ds_write_b32 traffic, 64 B/clk, fought the
//     other group's fragment reads and the phases stayed additive): the MFMA operand roles are swapped
//     (A = weights, B = pixels), so D[channel][pixel] puts pixel lane&31 on the lane and 4 consecutive channels
//     8*(reg>>2) + 4*(lane>>5) .. +3 in each register quad; one v_permlane32_swap per register exchanges quads
//     between the two lane halves, after which every lane holds 8 CONSECUTIVE channels of its pixel; after bias,
//     activation, residual and f16 packing a v_permlane16_swap per packed dword regroups the lanes so that one
//     16-B store instruction writes 64 contiguous bytes per pixel (16 cache lines per instruction instead of 32:
//     store / residual-load issue was the longest part of the E phase).  Residuals are loaded in that same layout.  The bias is read from a
//     small LDS table straight into the row-0 accumulators at the start of M0 and is the C operand of the first MFMAs.
#include <cstdlib>
#include "conv_common.h"

using namespace dsvc_conv;

#ifdef DSVC_PP_PROBE
// Build with DSVC_BUILD_DEFINES=-DDSVC_PP_PROBE to get, for wave 0 of each group of every workgroup, the shader-clock time
// between barriers: [g*8 + 2*i] = work of the group's phase i (0: M0, 1: S1, 2: M1, 3: E+S0), [g*8 + 2*i + 1] = wait at the barrier
// that ends it; [16] = iterations.  Stamps sit immediately before / after the workgroup barriers, where the LDS queue is
// drained anyway (tools/pp_phase_probe.py).
__device__ unsigned long long g_pp_phase[17];
extern "C" int dsvc_dbg_pp_phase_read(unsigned long long* out17, int reset) {
    if (hipMemcpyFromSymbol(out17, HIP_SYMBOL(g_pp_phase), sizeof(g_pp_phase)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[17] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pp_phase), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define PP_STAMP(v) { v = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#define PP_BARRIER(i) { unsigned long long tb_, ta_; PP_STAMP(tb_) lds_barrier(); PP_STAMP(ta_) ppw[i] += tb_ - t_last; ppb[i] += ta_ - tb_; t_last = ta_; }
#else
#define PP_BARRIER(i) lds_barrier();
#endif

#ifdef DSVC_PP_CLOCK
// Diagnostic build (DSVC_BUILD_DEFINES=-DDSVC_PP_CLOCK, tools/pp_clock.py): the in-kernel shader clock the guide's way
// (MI355X_MICROARCH.md, DVFS give-back item 6): wave 0 of group 0 of every workgroup stamps s_memtime (shader cycles) and
// s_memrealtime (100 MHz) once before and once after the tile loop; [0] += d(memtime), [1] += d(memrealtime), [2] += 1.
// The stamps go to a buffer nothing else reads; no output depends on them; the product build contains none of this.
__device__ unsigned long long g_pp_clock[3];
extern "C" int dsvc_dbg_pp_clock_read(unsigned long long* out3, int reset) {
    if (hipMemcpyFromSymbol(out3, HIP_SYMBOL(g_pp_clock), sizeof(g_pp_clock)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[3] = {0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pp_clock), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

namespace {

constexpr int PP_TH = 8, PP_PW = 34, PP_NPATCH = (PP_TH + 2) * PP_PW;   // 340 patch pixels
constexpr int PP_PFV = (PP_NPATCH * 4 + 255) / 256;                     // 6 prefetch vectors per thread
#ifndef PP_PF0_IN_S1
#define PP_PF0_IN_S1 1
#endif
constexpr bool PF0_IN_S1 = PP_PF0_IN_S1 != 0;
#ifndef PP_MIN_BLOCKS
#define PP_MIN_BLOCKS 2
#endif
#ifndef PP_VMEM_PRIO
#define PP_VMEM_PRIO 0
#endif
#ifndef PP_M_PRIO
#define PP_M_PRIO 3
#endif
#ifndef PP_RES_SPLIT
#define PP_RES_SPLIT 0
#endif
#ifndef PP_RES_IN_S1
#define PP_RES_IN_S1 0
#endif
#ifndef PP_POST_IN_M1       // fused 1x1 tail: its GEMM at the end of the wave's own M1 phase (1) or in the E phase (0: rounds 2-4)
#define PP_POST_IN_M1 1
#endif
constexpr bool POST_IN_M1 = PP_POST_IN_M1 != 0;
#ifndef PP_PATCH_SOFF       // 1: the tile enters the patch loads as the scalar offset operand (guarded descriptor); 0: added to the vector offset
#define PP_PATCH_SOFF 1
#endif
#ifndef PP_EPI_SOFF         // the same for the epilogue's stores / residual loads
#define PP_EPI_SOFF 1
#endif
// STORES keep the tile offset in the VECTOR offset (one v_add per store).  With the tile offset in the scalar-offset operand (a register)
// the frame was no longer reproducible: decoded frames differed from run to run by up to 1.0 in millions of elements -- while every launch
// in isolation, and every synchronised launch inside the frame, was bit-exact (tools/frame_determinism.py; round 5).  Bisected to exactly
// this: residual LOADS with a register scalar offset are fine, wide STORES are not.  The generated code issues
//     buffer_store_dwordx4 v[52:55], v56, s[16:19], s4 offen
//     v_or_b32 v52, ...                      <- overwrites the store's first data register in the next issue slot
// back to back: for a wide store WITHOUT a register offset the compiler inserts a wait state here (GCNHazardRecognizer: the 12-dword store
// hazard is taken to exist only when soffset is not a register); on gfx950 the data registers are evidently still being read in the
// register-offset case too when the memory pipeline is backed up.  With two wait states pinned behind every such store (-DPP_ST_NOP,
// PP_EPI_ST_SOFF=1) the frame is reproducible again [see profiles/r05_pp_epilogue_experiments.txt].  deepsvc_amd/build.py refuses any object that contains a
// wide buffer store with a register scalar offset.
#ifndef PP_EPI_ST_SOFF
#define PP_EPI_ST_SOFF 0
#endif
#ifndef PP_EPI_LD_SOFF
#define PP_EPI_LD_SOFF PP_EPI_SOFF
#endif
constexpr bool RES_IN_S1 = PP_RES_IN_S1 != 0;

__device__ __forceinline__ void buf_store16(rsrc_t r, unsigned off, uint4 v) {
    v4u_t t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_amdgcn_raw_buffer_store_b128(t, r, off, 0, 0);
}

// max(x, x*slope) for finite x in two VALU ops (fmaxf adds a canonicalising v_max per operand): med3(a, b, +inf) == max(a, b)
__device__ __forceinline__ float lrelu_f(float x, float slope) {
    return __builtin_amdgcn_fmed3f(x, x * slope, __builtin_inff());
}
// 8 packed f16 -> act(x) = max(x, x*slope) -> 8 packed f16 (slope 0: ReLU; 0<slope<1: LeakyReLU)
__device__ __forceinline__ uint4 act_h16x8(uint4 val, float slope) {
    unsigned u[4] = {val.x, val.y, val.z, val.w};
    if (slope == 0.f) {   // ReLU (wave-uniform): a f16 is negative iff its int16 image is, so one packed integer max per pair
#pragma unroll            // (-0.0 -> +0.0, NaNs with the sign bit set -> 0: neither occurs on this path)
        for (int k = 0; k < 4; ++k) asm("v_pk_max_i16 %0, %1, 0" : "=v"(u[k]) : "v"(u[k]));
        return make_uint4(u[0], u[1], u[2], u[3]);
    }
    // LeakyReLU on packed f16 in three VALU ops per pair instead of seven (2 cvt + 2 mul + 2 max + cvt_pk): the mixed-precision FMA
    // takes the f16 half, multiplies by the fp32 slope and rounds the exact product to f16 once -- the same value as cvt -> mul -> cvt --
    // and max commutes with that monotonic rounding, so max(x, f16(x * slope)) == f16(max(x, x * slope)) bit for bit.
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned t;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(t) : "v"(u[k]), "v"(slope));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(t) : "v"(u[k]), "v"(slope));
        asm("v_pk_max_f16 %0, %1, %2" : "=v"(u[k]) : "v"(u[k]), "v"(t));
    }
    return make_uint4(u[0], u[1], u[2], u[3]);
}

// RMIX (round 5): the residual add and the f16 packing of the epilogue as mixed-precision FMAs.  f16(x + r) for an fp32 x and a packed f16
// residual r took two v_cvt_f32_f16, two v_add_f32 and one v_cvt_pk_f16_f32 per output dword; v_fma_mixlo_f16 / v_fma_mixhi_f16 compute
// r.half * 1.0 + x in fp32 and round ONCE to the f16 half of the destination: two instructions per dword.  An E-phase instruction costs
// 15-25 clocks of issue beside the partner group's prioritised MFMA stream (in-kernel stamps), so the 96 instructions saved per
// tile and wave are the difference between E ~ 4800 and E ~ M.  Numerically the sum is now rounded once (fp32 sum -> f16) instead of twice
// (fp32 add, then f16): results can differ from the two-step form in the last f16 bit where the fp32-rounded sum sat exactly on an f16 tie.
__device__ __forceinline__ unsigned mix_pack_add(unsigned r, float x0, float x1) {      // {f16(r.lo + x0), f16(r.hi + x1)}
    unsigned t;                          // mixlo writes the low half, mixhi the high half: no zeroing move
    asm("v_fma_mixlo_f16 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(t) : "v"(r), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(t) : "v"(r), "v"(x1));
    return t;
}
__device__ __forceinline__ float mix_add_lo(unsigned r, float x) {      // fp32(r.lo + x): the same bits as x + float(r.lo)
    float t;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(t) : "v"(r), "v"(x));
    return t;
}
__device__ __forceinline__ float mix_add_hi(unsigned r, float x) {
    float t;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t) : "v"(r), "v"(x));
    return t;
}

// NCH: 32-channel input chunks (2: the schedule above; 1: inputs of <= 32 channels, phases M0 | E+S0 only).
// HEAD: narrow output head (Cout <= 8, any activation, fp32 or f16 out, optional residual of either type): one N-tile
// is computed and the lanes that hold channels 0..7 store them element-wise.
// POST: a 1x1 conv NT*32 -> NT*32 (ResBottleneckBlock.conv3) runs on the activated tile before the residual and the store: after
// the first swap level lane (r, h) holds, as packed f16, channels 16*kb + 8h .. +7 of pixel r for kb = 0..3 -- exactly the
// B-operand fragment of k-block kb of a 32x32x16 MFMA -- so the tile feeds 2 x 4 MFMAs per row against weight fragments
// read from LDS and the intermediate never leaves registers (it is rounded to f16 as a stored tensor would be).
// M16: the same schedule on v_mfma_f32_16x16x32_f16 (one MFMA per 64-byte chunk row and 16x16 output block) instead of
// v_mfma_f32_32x32x16_f16.  Same FLOP per cycle and the same LDS read count per FLOP; the chip holds a higher clock on this shape in
// MFMA + LDS loops (MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.14x at equal cycles), the epilogue needs one swap level
// instead of two and the residual arrives in the lanes' final layout.  Lane roles (j = lane & 15, q = lane >> 4):
//   operands: A = weights, row (output channel) 16*cb + j;  B = pixels, column (pixel) 16*half + rho(j) of tile row m;  both read
//     the 16-byte slot sigma(q) = {0, 2, 1, 3}[q] of their 80-byte LDS row, i.e. k-block q = input channels 8*sigma(q) .. +7;
//     rho = {1,3,5,7, 0,2,...,14, 9,11,13,15}: the lane -> row permutation (weight rows are stored permuted, pixels are simply
//     assigned permuted) that makes every 16-lane group of ds_read_b128 cover the 16 bank slots exactly once at the 80-byte pitch
//     for ANY base row (rows 5*row + sigma mod 16: odd rows with sigma in {0,1}, even rows with sigma in {2,3});
//   accumulators: lane (j, q) holds channels 16*cb + 4q .. +3 of pixel rho(j); one v_permlane16_swap per register between the
//     blocks cb = 2G and 2G + 1 leaves every lane with the 8 CONSECUTIVE channels 32G + 8*sigma(q) .. +7 -- 16 bytes of the pixel's
//     64-byte channel group, at byte offset 16*sigma(q): four lanes write 64 contiguous bytes, and the packed tile is at the same
//     time the B operand (k-block q) of the fused 1x1 conv.
//
// QC ("quad-contiguous" epilogue, round 5): in the layouts above the four lanes that cover the 64 contiguous bytes of a pixel are 16 lanes
// apart (the MFMA puts the pixel index in the LOW lane bits), so the four ADJACENT lanes of every quad of a store / residual-load
// instruction touch four different cache lines.  The texture addresser works through a 16-byte-per-lane instruction one quad per clock and
// merges only what a quad holds: 64 separate 16-byte accesses instead of 16 accesses of 64 bytes -- 64 clocks of the CU's one address path per
// instruction instead of 16.  That fits the measured E phases: (8 stores + 8 loads per residual operand) x 64 clocks x the 4 waves of a
// group = 2048 / 4096 / 6144 clocks against 2190-2900 / 4730-4800 / 7690-8190 measured for 0 / 1 / 2 residual operands.
// With QC the packed vectors change lanes once more on their way out -- lane 4*pixel + piece, one ds_permute_b32 per dword: the LDS
// crossbar, no LDS memory -- so that a quad writes 64 contiguous bytes; residual vectors are loaded in that layout and pulled back with
// ds_bpermute_b32.  Pure data movement: results are bit-identical to QC = false (tests/test_gpu_ops.py).
template <int NT, int NCH, bool HEAD, bool PRE, bool ACT, bool HAS_R1, bool HAS_R2, bool POST = false, bool M16 = false, bool QC = false, bool RMIX = false>
__global__ void __launch_bounds__(512, M16 ? 1 : PP_MIN_BLOCKS) conv3x3_pp_kernel(const ConvParams p, int ntiles, int iters, float act_slope,
                                                              float pre_slope) {
    DSVC_F16_SAT();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTN = NT * 32, MTW = 2;
    // where the next tile's chunk-0 loads are issued: in the S1 phase (the shortest one) instead of the E phase (the longest: 8 stores
    // and, with a residual, 8 more loads -- a vector-memory instruction costs 150-250 cycles to issue beside the partner group's
    // MFMA + LDS stream).  Same-box A/B on an idle GPU (tools/bench_pp.py): + residual 212 -> 194 us, two residuals 262 -> 243,
    // fused tail 232 -> 218, plain 193 -> 200; inside the frame (bench.py --lanes 1, rocprofv3): + residual 218 -> 191 us, fused
    // tail 142 -> 124, PMC traffic of the residual variants 1.24x -> 1.07x algorithmic, all variants moved: another -1 % on the
    // launch average.  NCH == 1 variants have no S1 phase.
    constexpr bool PF0S1 = PF0_IN_S1 && NCH == 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PP_GROUP_PARITY
    const int g = wave & 1, wg = wave >> 1, gtid = (wg << 6) | lane;   // group, wave in group, thread in group
#else
    const int g = wave >> 2, wg = wave & 3, gtid = tid & 255;      // group, wave in group, thread in group
#endif
    const int r = lane & 31, h = lane >> 5;
    static_assert(!(M16 && HEAD), "narrow heads stay on the 32x32x16 shape");
    const int j16 = lane & 15, q16 = lane >> 4;
    const int rho16 = j16 < 4 ? 2 * j16 + 1 : (j16 < 12 ? 2 * (j16 - 4) : 2 * (j16 - 12) + 9);
    const int sig16 = q16 == 1 ? 2 : (q16 == 2 ? 1 : q16);
    // QC: byte address (lane x 4) of the lane that holds THIS lane's (pixel, 16-byte piece) in the quad-contiguous layout lane = 4*pixel + piece
    const int qc_addr = 4 * (M16 ? 4 * rho16 + sig16 : 4 * (lane & 15) + 2 * ((lane >> 4) & 1) + (lane >> 5));
    auto to_qc = [&](uint4 v) -> uint4 {        // epilogue layout -> quad-contiguous (push)
        if constexpr (!QC) return v;
        return make_uint4((unsigned)__builtin_amdgcn_ds_permute(qc_addr, (int)v.x), (unsigned)__builtin_amdgcn_ds_permute(qc_addr, (int)v.y),
                          (unsigned)__builtin_amdgcn_ds_permute(qc_addr, (int)v.z), (unsigned)__builtin_amdgcn_ds_permute(qc_addr, (int)v.w));
    };
    auto from_qc = [&](uint4 v) -> uint4 {      // quad-contiguous -> epilogue layout (pull)
        if constexpr (!QC) return v;
        return make_uint4((unsigned)__builtin_amdgcn_ds_bpermute(qc_addr, (int)v.x), (unsigned)__builtin_amdgcn_ds_bpermute(qc_addr, (int)v.y),
                          (unsigned)__builtin_amdgcn_ds_bpermute(qc_addr, (int)v.z), (unsigned)__builtin_amdgcn_ds_bpermute(qc_addr, (int)v.w));
    };
    auto perm_row = [](int row) {               // M16: LDS position of weight row `row` inside its 16-row block (rho of the block index)
        if constexpr (!M16) return row;
        const int i = row & 15;
        return (row & ~15) | (i < 4 ? 2 * i + 1 : (i < 12 ? 2 * (i - 4) : 2 * (i - 12) + 9));
    };
    char* lds_w = smem;                                                       // [chunk 2][tap 9][NTN] rows of 80 B
    char* lds_a = smem + NCH * 9 * NTN * ROWB + g * (PP_NPATCH * ROWB);        // this group's patch
    float* lds_bias = (float*)(smem + NCH * 9 * NTN * ROWB + 2 * PP_NPATCH * ROWB);   // NTN floats (+ NTN of the 1x1's bias)
    char* lds_w3 = (char*)(lds_bias + 2 * NTN);                                        // POST: [chunk NT][NTN] rows of 80 B
    const rsrc_t rout = make_rsrc(p.out, p.out_bytes);
    const rsrc_t rr1 = make_rsrc(HAS_R1 ? p.res1 : p.out, HAS_R1 ? p.res1_bytes : p.out_bytes);
    const rsrc_t rr2 = make_rsrc(HAS_R2 ? p.res2 : p.out, HAS_R2 ? p.res2_bytes : p.out_bytes);

    const int H = p.H, W = p.W, tiles_x = p.tiles_x, tiles_y = p.tiles_y;
    // Address arithmetic is kept off the vector ALU as far as possible (it runs beside the other group's MFMA phase):
    // tile coordinates are wave-uniform and computed once per iteration; per-thread patch offsets are loop-invariant;
    // products use the full-rate 24-bit multiplier (B*H*W < 2^24 is an eligibility condition).
    struct TileOrg { int b, oy0, ox0; };
    auto tile_origin = [&](int tile) -> TileOrg {
        const int tx = tile % tiles_x;
        const int rest = tile / tiles_x;
        const int ty = rest % tiles_y;
        return TileOrg{rest / tiles_y, ty * PP_TH, tx * 32};
    };
    // patch vector v = gtid + 256*j -> pixel v>>2 (row-major in the 10x34 patch), 16-B quarter v&3
    // [r5] Addressing diet: an S / E-phase instruction costs 13-20 clocks of issue beside the partner group's prioritised MFMA stream and every
    // one of them also delays that stream, and the old code spent ~6 vector instructions per load on the address and the border mask plus two
    // scalar loads of the segment descriptor per chunk and iteration.  Now everything that does not depend on the tile is computed once:
    // per chunk a buffer descriptor, and per thread and patch vector the byte offset inside the patch pvo[c][j] = patch pixel x pitch +
    // quarter (| OOB where the vector is statically absent: short chunk, tail of the last vector); the tile enters as the SCALAR offset
    // operand of the load (tile origin x pitch + chunk), and the border mask is computed only for tiles that touch the image border (a
    // wave-uniform branch around the mask arithmetic alone: no load sits inside it).  The descriptors start one image row + one pixel
    // BEFORE the tensor (guard), so the scalar offset of the top-left tile, whose halo starts at pixel (-1, -1), is not negative; the
    // lanes that would read the guard are exactly the masked padding lanes.
    const int q = gtid & 3;
    const bool last_ok = (gtid >> 2) + 64 * (PP_PFV - 1) < PP_NPATCH;      // only the last vector can fall off the patch
    rsrc_t rin_c[NCH];
    unsigned ldb_c[NCH], sbase_c[NCH], pvo[NCH][PP_PFV];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const ConvSeg& sg = p.seg[(p.nseg > 1 && c >= p.seg[1].chunk0) ? 1 : 0];
        const unsigned ldb = (unsigned)sg.ld * 2, guard = PP_PATCH_SOFF ? (unsigned)(W + 1) * ldb : 0u;
        rin_c[c] = make_rsrc((const char*)sg.ptr - guard, sg.bytes + guard);
        ldb_c[c] = ldb;
        sbase_c[c] = guard + (unsigned)(c - sg.chunk0) * 64;
        const bool cok = (c - sg.chunk0) * 32 + q * 8 < sg.C;              // short chunk (C % 8 == 0): the tail reads zeros
#pragma unroll
        for (int j = 0; j < PP_PFV; ++j) {
            const int pix = (gtid >> 2) + 64 * j, py = pix / PP_PW, px = pix - py * PP_PW;
            pvo[c][j] = ((unsigned)(py * W + px) * ldb + q * 16) | ((cok && (j < PP_PFV - 1 || last_ok)) ? 0u : OOB);
        }
    }
    auto load_chunk = [&](const TileOrg& t, const int ch, uint4 (&pf)[PP_PFV]) {   // ch: a literal at every call site
        const int base_pix = (t.b * H + t.oy0 - 1) * W + t.ox0 - 1;        // scalar; >= -(W + 1): the guard makes the byte offset non-negative
        const unsigned so = (unsigned)(base_pix * (int)ldb_c[ch]) + sbase_c[ch];
        const bool interior = t.oy0 >= 1 && t.oy0 + PP_TH + 1 <= H && t.ox0 >= 1 && t.ox0 + PP_PW - 1 <= W;   // wave-uniform
        unsigned dm[PP_PFV];
#pragma unroll
        for (int j = 0; j < PP_PFV; ++j) dm[j] = 0u;
        if (!interior) {                                                     // image border: padding reads zeros
            int gt = gtid;
            asm volatile("" : "+v"(gt));                                     // pins the mask arithmetic INSIDE the branch (it had been hoisted: 24 vector instructions per chunk on every tile)
#pragma unroll
            for (int j = 0; j < PP_PFV; ++j) {
                const int pix = (gt >> 2) + 64 * j, py = pix / PP_PW, px = pix - py * PP_PW;
                const unsigned iy = (unsigned)(t.oy0 - 1 + py), ix = (unsigned)(t.ox0 - 1 + px);
                dm[j] = (iy < (unsigned)H && ix < (unsigned)W) ? 0u : OOB;
            }
        }
        if (PP_VMEM_PRIO) __builtin_amdgcn_s_setprio(PP_VMEM_PRIO);
#pragma unroll
        for (int j = 0; j < PP_PFV; ++j) {
            const v4u_t v = PP_PATCH_SOFF ? __builtin_amdgcn_raw_buffer_load_b128(rin_c[ch], pvo[ch][j] | dm[j], so, 0)
                                          : __builtin_amdgcn_raw_buffer_load_b128(rin_c[ch], ((pvo[ch][j] + so) & 0x7FFFFFFFu) | (pvo[ch][j] & OOB) | dm[j], 0, 0);
            pf[j] = make_uint4(v.x, v.y, v.z, v.w);
        }
        if (PP_VMEM_PRIO) __builtin_amdgcn_s_setprio(0);
    };
    auto store_chunk = [&](uint4 (&pf)[PP_PFV]) {
#pragma unroll
        for (int j = 0; j < PP_PFV; ++j) {
            const int pix = (gtid >> 2) + 64 * j;
            if (pix >= PP_NPATCH) continue;
            *(uint4*)(lds_a + pix * ROWB + q * 16) = PRE ? act_h16x8(pf[j], pre_slope) : pf[j];
        }
    };

    int a_off[MTW], b_off[NT];
#pragma unroll
    for (int m = 0; m < MTW; ++m) a_off[m] = ((wg * MTW + m) * PP_PW + r) * ROWB + h * 32;
#pragma unroll
    for (int n = 0; n < NT; ++n) b_off[n] = (n * 32 + r) * ROWB + h * 32;

    f32x16 acc[MTW][NT];
    // 18 k-steps (9 taps x 2 halves of the 64-B chunk); the fragments of step s+1 are read before the MFMAs of step s
    auto mfma_phase = [&](int ch, const bool FIRST) {   // FIRST (chunk 0, a literal at both call sites): start from the bias
        const char* wt = lds_w + (size_t)ch * 9 * NTN * ROWB;
        auto read_frags = [&](int s, uint4 (&af)[MTW], uint4 (&bf)[NT]) {
            const int t = s >> 1, ks = s & 1;
            const int tap_off = ((t / 3) * PP_PW + (t % 3)) * ROWB;
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[n] = *(const uint4*)(wt + t * NTN * ROWB + b_off[n] + ks * 16);
#pragma unroll
            for (int m = 0; m < MTW; ++m) af[m] = *(const uint4*)(lds_a + a_off[m] + tap_off + ks * 16);
        };
        if (FIRST) {   // chunk 0: row-0 accumulators start as the bias (reg i <-> channel (i&3) + 8*(i>>2) + 4*h);
#pragma unroll               // row 1 takes them as the C operand of its first MFMA, issued before row 0's
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float4 bq = *(const float4*)(lds_bias + n * 32 + 8 * qd + 4 * h);
                    acc[0][n][4 * qd] = bq.x; acc[0][n][4 * qd + 1] = bq.y; acc[0][n][4 * qd + 2] = bq.z; acc[0][n][4 * qd + 3] = bq.w;
                }
        }
        uint4 af[3][MTW], bf[3][NT];                 // fragments are read TWO k-steps ahead: one step (4 MFMAs = 128 cycles)
        read_frags(0, af[0], bf[0]);                // does not cover the ds_read latency while the other group also uses LDS
        read_frags(1, af[1], bf[1]);
        __builtin_amdgcn_s_setprio(PP_M_PRIO);      // the SIMD's issue arbiter otherwise favours the older group-0 waves:
#pragma unroll                                      // their VALU-dense epilogue starved group 1's MFMAs (in-kernel stamps)
        for (int s2 = 0; s2 < 18; ++s2) {
            if (s2 + 2 < 18) read_frags(s2 + 2, af[(s2 + 2) % 3], bf[(s2 + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mm = 0; mm < MTW; ++mm)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int m = MTW - 1 - mm;      // row 1 first: its first MFMA reads row 0's still-untouched bias
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, bf[s2 % 3][n]),   // D[channel][pixel]
                                                                         __builtin_bit_cast(h16x8, af[s2 % 3][m]),
                                                                         (FIRST && s2 == 0) ? acc[0][n] : acc[m][n], 0, 0, 0);
                }
            // Empty asm that "updates" every accumulator: a data + chain dependency per k-step.  The MFMA intrinsics are
            // pure, so instruction selection is otherwise free to sink a whole accumulation chain (18 MFMAs, all their
            // fragments live -> spills) towards its only user, the epilogue; sched_barrier binds only the later scheduler.
            if constexpr (NT == 2) asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
            else asm volatile("" : "+v"(acc[0][0]), "+v"(acc[1][0]));
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- M16: 16x16x32 shape ----
    constexpr int NCB = NT * 2;                 // 16-channel output blocks
    f32x4 acc16[4][NCB];                        // [pixel block pb = 2*m + half][channel block cb]
    int a16_off[NCB], b16_off[4];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) a16_off[cb] = (16 * cb + rho16) * ROWB + 16 * sig16;
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) b16_off[pb] = ((wg * MTW + (pb >> 1)) * PP_PW + 16 * (pb & 1) + rho16) * ROWB + 16 * sig16;
    auto mfma_phase16 = [&](int ch, const bool FIRST) {       // 9 taps x 4 x NCB MFMAs; the fragments of tap t+1 are read before the MFMAs of tap t
        const char* wt = lds_w + (size_t)ch * 9 * NTN * ROWB;
        auto read_tap = [&](int t, uint4 (&af)[NCB], uint4 (&bf)[4]) {
            const int tap_off = ((t / 3) * PP_PW + (t % 3)) * ROWB;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) af[cb] = *(const uint4*)(wt + t * NTN * ROWB + a16_off[cb]);
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) bf[pb] = *(const uint4*)(lds_a + b16_off[pb] + tap_off);
        };
        f32x4 bias4[NCB];
        if (FIRST) {                            // chunk 0 starts from the bias: the C operand of every block's first MFMA
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const float4 bq = *(const float4*)(lds_bias + 16 * cb + 4 * q16);
                bias4[cb][0] = bq.x; bias4[cb][1] = bq.y; bias4[cb][2] = bq.z; bias4[cb][3] = bq.w;
            }
        }
        uint4 af[2][NCB], bf[2][4];
        read_tap(0, af[0], bf[0]);
        __builtin_amdgcn_s_setprio(PP_M_PRIO);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) read_tap(t + 1, af[(t + 1) & 1], bf[(t + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int pb = 0; pb < 4; ++pb)
                    acc16[pb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, af[t & 1][cb]),      // D[channel][pixel]
                                                                            __builtin_bit_cast(h16x8, bf[t & 1][pb]),
                                                                            (FIRST && t == 0) ? bias4[cb] : acc16[pb][cb], 0, 0, 0);
            // pin the accumulation order (see mfma_phase)
            if constexpr (NT == 2) asm volatile("" : "+v"(acc16[0][0]), "+v"(acc16[0][1]), "+v"(acc16[0][2]), "+v"(acc16[0][3]),
                                                     "+v"(acc16[1][0]), "+v"(acc16[1][1]), "+v"(acc16[1][2]), "+v"(acc16[1][3]),
                                                     "+v"(acc16[2][0]), "+v"(acc16[2][1]), "+v"(acc16[2][2]), "+v"(acc16[2][3]),
                                                     "+v"(acc16[3][0]), "+v"(acc16[3][1]), "+v"(acc16[3][2]), "+v"(acc16[3][3]));
            else asm volatile("" : "+v"(acc16[0][0]), "+v"(acc16[0][1]), "+v"(acc16[1][0]), "+v"(acc16[1][1]),
                                   "+v"(acc16[2][0]), "+v"(acc16[2][1]), "+v"(acc16[3][0]), "+v"(acc16[3][1]));
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // Epilogue addressing [r5: the same diet as the patch loads].  MEMORY layout of a tile row after the swaps: 32x32x16: lane l of vector
    // `half` holds pixel (l&15) + 16*half, channels 16*((l>>4)&1) + 8*(l>>5) .. +7 of the N-tile (4 lanes cover 64 contiguous bytes of a
    // pixel, one instruction touches 16 cache lines); 16x16x32: lane (j, q) <-> pixel 16*half + rho(j), bytes 16*sigma(q) .. +15 of the
    // 64-byte channel group.  Per tensor ONE invariant per-lane byte offset (pixel-in-half x pitch + piece); per tile row m and half a
    // SCALAR pixel index (x pitch = the scalar offset operand of the store / load) and a mask that is computed only for tiles that touch
    // the picture's right / bottom edge or have no tile at all.  Index e = 2*m + half (= the 16x16x32 path's pixel block pb).
    const int px_lane = QC ? lane >> 2 : (M16 ? rho16 : r & 15);
    const unsigned piece_b = QC ? 16u * (lane & 3) : (M16 ? 16u * sig16 : 32u * (r >> 4) + 16u * h);
    const unsigned ldo2 = (unsigned)p.out_ld * 2, ld12 = (unsigned)p.res1_ld * 2, ld22 = (unsigned)p.res2_ld * 2;
    const unsigned evo_out = (unsigned)px_lane * ldo2 + piece_b, evo_r1 = (unsigned)px_lane * ld12 + piece_b, evo_r2 = (unsigned)px_lane * ld22 + piece_b;
    struct EpiRows { unsigned spix[2 * MTW], dm[2 * MTW]; };
    auto epi_rows = [&](const TileOrg& t, bool valid, EpiRows& er) {
        const bool interior = valid && t.oy0 + PP_TH <= p.Ho && t.ox0 + 32 <= p.Wo;          // wave-uniform
#pragma unroll
        for (int e = 0; e < 2 * MTW; ++e) {
            const int oy = t.oy0 + wg * MTW + (e >> 1);                                         // scalar
            er.spix[e] = (unsigned)((t.b * p.Ho + oy) * p.Wo + t.ox0 + 16 * (e & 1));
            er.dm[e] = 0u;
        }
        if (!interior) {
            int pl = px_lane;
            asm volatile("" : "+v"(pl));                                                           // keeps the mask arithmetic inside the branch
#pragma unroll
            for (int e = 0; e < 2 * MTW; ++e) {
                const int oy = t.oy0 + wg * MTW + (e >> 1);
                er.dm[e] = (valid && oy < p.Ho && t.ox0 + 16 * (e & 1) + pl < p.Wo) ? 0u : OOB;
            }
        }
    };
    auto ld16 = [&](rsrc_t rs, unsigned voff, unsigned soff) -> uint4 {
        const v4u_t v = PP_EPI_LD_SOFF ? __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0)
                                    : __builtin_amdgcn_raw_buffer_load_b128(rs, (((voff & 0x7FFFFFFFu) + soff) & 0x7FFFFFFFu) | (voff & OOB), 0, 0);
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    auto st16 = [&](rsrc_t rs, unsigned voff, unsigned soff, uint4 v) {
        v4u_t t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        if (PP_EPI_ST_SOFF) {
            __builtin_amdgcn_raw_buffer_store_b128(t, rs, voff, soff, 0);
#ifdef PP_ST_NOP      // experiment: two wait states pinned right behind a store that has a REGISTER scalar offset (see PP_EPI_ST_SOFF)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 1");
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        else __builtin_amdgcn_raw_buffer_store_b128(t, rs, (((voff & 0x7FFFFFFFu) + soff) & 0x7FFFFFFFu) | (voff & OOB), 0, 0);
    };
    uint4 res[MTW * NT * 2];
    auto load_res = [&](const TileOrg& t, bool valid, int m0 = 0, int m1 = MTW) {
        EpiRows er;
        epi_rows(t, valid, er);
#pragma unroll
        for (int e = 2 * m0; e < 2 * m1; ++e)
#pragma unroll
            for (int n = 0; n < NT; ++n)          // M16: res[pb * NT + G]; 32x32x16: res[(m * NT + n) * 2 + half]
                res[M16 ? e * NT + n : ((e >> 1) * NT + n) * 2 + (e & 1)] = ld16(rr1, (evo_r1 | er.dm[e]) + 64 * n, er.spix[e] * ld12);
    };

    // ---- POST: the fused 1x1 conv (ResBottleneckBlock.conv3) on the activated tile ----
    // [r5] It runs at the END of the wave's own M1 phase (PP_POST_IN_M1), not in the E phase: there its 16 MFMAs per wave queued behind the
    // partner group's prioritised MFMA stream and the E phase took 6000-7000 clocks against 3000 for M (in-kernel stamps,
    // profiles/r05_pp_phase_probe.txt); in M1 the wave owns the matrix pipe anyway and its partner is in the short S1 phase.
    static_assert(!POST || NT == 2, "fused 1x1: 64 -> 64 channels");
    auto post_gemm = [&]() {
        if constexpr (POST && M16) {
            __builtin_amdgcn_s_setprio(PP_M_PRIO);
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                uint4 bfr[2];                // B fragments of the 1x1's two k-groups: act(conv3x3 + bias), packed f16
#pragma unroll
                for (int G = 0; G < 2; ++G) {
                    float lo[4], hi[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {       // one v_permlane16_swap per register between the channel blocks 2G and 2G + 1 (see the epilogue)
                        const float xq = acc16[pb][2 * G][v], yq = acc16[pb][2 * G + 1][v];
                        const v2u_t sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, xq), __builtin_bit_cast(unsigned, yq), false, false);
                        const unsigned lo_ = sw.x, hi_ = sw.y;
                        lo[v] = __builtin_bit_cast(float, lo_); hi[v] = __builtin_bit_cast(float, hi_);
                    }
                    if constexpr (ACT) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) { lo[v] = lrelu_f(lo[v], act_slope); hi[v] = lrelu_f(hi[v], act_slope); }
                    }
                    bfr[G] = make_uint4(pk_h16(lo[0], lo[1]), pk_h16(lo[2], lo[3]), pk_h16(hi[0], hi[1]), pk_h16(hi[2], hi[3]));
                }
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {     // accumulators of the 1x1 start as its bias
                    const float4 bq = *(const float4*)(lds_bias + NTN + 16 * cb + 4 * q16);
                    acc16[pb][cb][0] = bq.x; acc16[pb][cb][1] = bq.y; acc16[pb][cb][2] = bq.z; acc16[pb][cb][3] = bq.w;
                }
#pragma unroll
                for (int G = 0; G < 2; ++G)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        const uint4 wf = *(const uint4*)(lds_w3 + (G * NTN + 16 * cb + rho16) * ROWB + 16 * sig16);
                        acc16[pb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, wf), __builtin_bit_cast(h16x8, bfr[G]), acc16[pb][cb], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_s_setprio(0);
        } else if constexpr (POST) {
            __builtin_amdgcn_s_setprio(PP_M_PRIO);
#pragma unroll
            for (int M = 0; M < MTW; ++M) {
                unsigned bfr[4][4];          // B fragments of k-blocks 0..3: act(conv3x3 + bias) of pixel r, packed f16
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int P = 0; P < 2; ++P) {
                        float v[8];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float xq = acc[M][n][8 * P + j], yq = acc[M][n][8 * P + 4 + j];
                            const v2u_t sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xq), __builtin_bit_cast(unsigned, yq), false, false);
                            const unsigned lo_ = sw.x, hi_ = sw.y;
                            v[j] = __builtin_bit_cast(float, lo_); v[4 + j] = __builtin_bit_cast(float, hi_);
                        }
                        if constexpr (ACT) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = lrelu_f(v[j], act_slope);
                        }
#pragma unroll
                        for (int d = 0; d < 4; ++d) bfr[2 * n + P][d] = pk_h16(v[2 * d], v[2 * d + 1]);
                    }
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2)       // accumulators of the 1x1 start as its bias
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const float4 bq = *(const float4*)(lds_bias + NTN + n2 * 32 + 8 * qd + 4 * h);
                        acc[M][n2][4 * qd] = bq.x; acc[M][n2][4 * qd + 1] = bq.y; acc[M][n2][4 * qd + 2] = bq.z; acc[M][n2][4 * qd + 3] = bq.w;
                    }
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int n2 = 0; n2 < 2; ++n2) {
                        const uint4 wf = *(const uint4*)(lds_w3 + ((kb >> 1) * NTN + n2 * 32 + r) * ROWB + (kb & 1) * 32 + h * 16);
                        const uint4 bf4 = make_uint4(bfr[kb][0], bfr[kb][1], bfr[kb][2], bfr[kb][3]);
                        acc[M][n2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, wf), __builtin_bit_cast(h16x8, bf4), acc[M][n2], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_s_setprio(0);
        }
    };

    // XCD-aware walk: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so XCD x = blockIdx % 8
    // takes the x-th contiguous eighth of the tile raster (a band of tile rows) and its 32 workgroups x 2 groups sweep
    // that band ~one tile row per iteration: the halo rows shared by vertically adjacent tiles are then re-read from
    // the SAME L2 one iteration later (PMC: input fetch was 1.38x algorithmic with a chip-wide raster walk).
    const int nband = (gridDim.x % 8 == 0) ? 8 : 1, per = (int)gridDim.x / nband;
    const int band = (int)blockIdx.x % nband, band_tiles = (ntiles + nband - 1) / nband;
    const int lt0 = ((int)blockIdx.x / nband) * 2 + g;
    auto tile_raw = [&](int k) {           // tile index, or ntiles when this (group, k) has no tile
        const int lt = lt0 + k * 2 * per;
        const int t = band * band_tiles + lt;
        return (lt < band_tiles && t < ntiles) ? t : ntiles;
    };
    auto tile_of = [&](int k) { return tile_raw(k); };
    // All global loads are ISSUED in the S / E phases (which have slack) and consumed a full iteration (four phases,
    // ~7 us) later: each input-chunk stream has its own register buffer (pf0: chunk 0 of tile k+2, pf1: chunk 1 of
    // tile k+1) and the residual of tile k+1 is requested as soon as tile k's epilogue has consumed its registers.
    // In-kernel stamps showed a load issued ONE M phase (~2 us) ahead still stalling its consumer, and an M phase with
    // the prefetch address arithmetic in front of its MFMAs taking ~4000 cycles instead of ~2700.
    uint4 pf0[PP_PFV], pf1[PP_PFV];
    TileOrg cur = tile_origin(min(tile_of(0), ntiles - 1)), nxt = tile_origin(min(tile_of(1), ntiles - 1));
    load_chunk(cur, 0, pf0);
    // ---- weights: staged once per workgroup (rows are [chunk][tap][Np == NTN] x 64 B in global memory).  All loads of a thread are issued
    // back to back behind the first patch loads (bounds-checked buffer loads, masked by the offset: conditional loads had been compiled to
    // one branch + L2 round trip each), so the prologue costs ONE memory round trip instead of ~ten ----
    {
        constexpr int nrows = NCH * 9 * NTN, NWV = (nrows + 127) / 128;
        constexpr int nrows3 = POST ? NT * NTN : 0, NWV3 = (nrows3 + 127) / 128;
        const int q4 = tid & 3;
        const rsrc_t rw = make_rsrc(p.w, (unsigned)nrows * 64u);
        const rsrc_t rw3 = make_rsrc(POST ? p.post_w : p.w, (unsigned)(POST ? nrows3 : nrows) * 64u);
        uint4 wv[NWV], wv3[NWV3 > 0 ? NWV3 : 1];
#pragma unroll
        for (int j = 0; j < NWV; ++j) {
            const unsigned row = (unsigned)(tid >> 2) + 128u * j;
            wv[j] = buf_load16(rw, (row * 64u + q4 * 16u) | (row < (unsigned)nrows ? 0u : OOB));
        }
#pragma unroll
        for (int j = 0; j < NWV3; ++j) {
            const unsigned row = (unsigned)(tid >> 2) + 128u * j;
            wv3[j] = buf_load16(rw3, (row * 64u + q4 * 16u) | (row < (unsigned)nrows3 ? 0u : OOB));
        }
        if (tid < NTN) lds_bias[tid] = (p.bias && tid < p.Cout) ? p.bias[tid] : 0.f;
        if constexpr (POST) {
            if (tid < NTN) lds_bias[NTN + tid] = (p.post_bias && tid < p.Cout) ? p.post_bias[tid] : 0.f;
        }
        store_chunk(pf0);                      // the patch loads were issued first and return first
#pragma unroll
        for (int j = 0; j < NWV; ++j) {
            const int row = (tid >> 2) + 128 * j;
            if (row < nrows) *(uint4*)(lds_w + perm_row(row) * ROWB + q4 * 16) = wv[j];
        }
#pragma unroll
        for (int j = 0; j < NWV3; ++j) {
            const int row = (tid >> 2) + 128 * j;
            if (row < nrows3) *(uint4*)(lds_w3 + perm_row(row) * ROWB + q4 * 16) = wv3[j];
        }
    }
    if constexpr (NCH == 2) load_chunk(cur, 1, pf1);
    if constexpr (!PF0S1) load_chunk(nxt, 0, pf0);
    if constexpr (HAS_R1 && !(RES_IN_S1 && NCH == 2)) load_res(cur, tile_of(0) < ntiles, 0, (PP_RES_SPLIT && NCH == 2) ? 1 : MTW);
    lds_barrier();                       // weights, bias and both groups' first patches are in place
    if (g) lds_barrier();                // group 1 runs one phase behind group 0

#ifdef DSVC_PP_PROBE
    unsigned long long ppw[4] = {0, 0, 0, 0}, ppb[4] = {0, 0, 0, 0}, t_last;
    PP_STAMP(t_last)
#endif
#ifdef DSVC_PP_CLOCK
    const unsigned long long ck_t0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int k = 0; k < iters; ++k) {
        const bool valid = tile_of(k) < ntiles;
        // ---- M0 ----
        if constexpr (M16) mfma_phase16(0, true); else mfma_phase(0, true);
        PP_BARRIER(0)
        if constexpr (NCH == 2) {
            // ---- S1 (+ prefetch: next tile's chunk 1) ----
            store_chunk(pf1);
            load_chunk(nxt, 1, pf1);
            // the next tile's chunk 0 is requested HERE, not in the E phase: a vector-memory instruction costs 150-250 cycles to
            // issue beside the partner group's MFMA + LDS stream, and the E phase (8 stores, 8-12 residual loads) is the longest
            // phase while this one is the shortest; the data is consumed at the end of this iteration's E phase (M1 + E later)
            if constexpr (PF0S1) load_chunk(nxt, 0, pf0);
            if constexpr (HAS_R1 && RES_IN_S1) load_res(cur, valid);
            if constexpr (HAS_R1 && PP_RES_SPLIT && NCH == 2) load_res(cur, valid, 1, 2);
            PP_BARRIER(1)
            // ---- M1 ----
            if constexpr (M16) mfma_phase16(1, false); else mfma_phase(1, false);
            if constexpr (POST && POST_IN_M1) post_gemm();
            PP_BARRIER(2)
        }
        // ---- E + S0 (+ prefetch: chunk 0 of the tile after next, the next tile's residual) ----
        if constexpr (HEAD) {
            // narrow head: after the first swap level the h == 0 lanes hold channels 0..7 of pixel r (bias already in acc)
#define PP_HEAD_ROW(M)                                                                                              \
    {                                                                                                               \
        float v[8];                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                             \
            const float xq = acc[M][0][j], yq = acc[M][0][4 + j];                                                   \
            const v2u_t sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xq), __builtin_bit_cast(unsigned, yq), false, false); \
            const unsigned lo_ = sw.x, hi_ = sw.y;                                                                  \
            v[j] = __builtin_bit_cast(float, lo_); v[4 + j] = __builtin_bit_cast(float, hi_);                       \
        }                                                                                                           \
        const int oy = cur.oy0 + wg * MTW + M, ox = cur.ox0 + r;                                                    \
        if (valid && h == 0 && oy < p.Ho && ox < p.Wo) {                                                            \
            const int64_t opix = ((int64_t)cur.b * p.Ho + oy) * p.Wo + ox;                                          \
            _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                           \
                if (j < p.Cout) {                                                                                   \
                    float val = apply_act(v[j], p.act, p.slope);                                                    \
                    if (p.res1) val += load_elem(p.res1, p.res1_dtype, opix * p.res1_ld + j);                       \
                    store_elem(p.out, p.out_dtype, opix * p.out_ld + j, val);                                       \
                }                                                                                                   \
        }                                                                                                           \
    }
            PP_HEAD_ROW(0) PP_HEAD_ROW(1)
#undef PP_HEAD_ROW
        } else if constexpr (M16) {
            EpiRows er;
            epi_rows(cur, valid, er);
            // one v_permlane16_swap per register between the channel blocks 2G (X) and 2G + 1 (Y): X' = rows {X0, Y0, X2, Y2},
            // Y' = rows {X1, Y1, X3, Y3} -> lane row q holds channels 32G + 8*sigma(q) + v in X'[v] and ... + 4 + v in Y'[v]
#define SWAP16F(X, Y, LO, HI)                                                                                       \
    { const float xq = (X), yq = (Y);   /* scalars first: hipcc 7.2 mis-casts vector elements */                    \
      const v2u_t sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, xq), __builtin_bit_cast(unsigned, yq), false, false); \
      const unsigned lo_ = sw.x, hi_ = sw.y; (LO) = __builtin_bit_cast(float, lo_); (HI) = __builtin_bit_cast(float, hi_); }
#define PP16_GATHER(PB, G, LO, HI)                                                                                  \
    _Pragma("unroll") for (int v = 0; v < 4; ++v) SWAP16F(acc16[PB][2 * (G)][v], acc16[PB][2 * (G) + 1][v], LO[v], HI[v])
            if constexpr (POST && !POST_IN_M1) post_gemm();
#define PP16_OUT(PB, G)                                                                                             \
    if constexpr (G < NT) {                                                                                         \
        float lo[4], hi[4];                                                                                         \
        PP16_GATHER(PB, G, lo, hi)                                                                                  \
        if constexpr (ACT && !POST) {                                                                               \
            _Pragma("unroll") for (int v = 0; v < 4; ++v) { lo[v] = lrelu_f(lo[v], act_slope); hi[v] = lrelu_f(hi[v], act_slope); } \
        }                                                                                                           \
        if constexpr (HAS_R1 && RMIX) {                                                                             \
            uint4 ra = from_qc(res[PB * NT + G]);                                                                   \
            if constexpr (HAS_R2) {      /* (x + r1) in fp32 -- the bits of the two-step form --, then f16(. + r2) */  \
                lo[0] = mix_add_lo(ra.x, lo[0]); lo[1] = mix_add_hi(ra.x, lo[1]); lo[2] = mix_add_lo(ra.y, lo[2]); lo[3] = mix_add_hi(ra.y, lo[3]); \
                hi[0] = mix_add_lo(ra.z, hi[0]); hi[1] = mix_add_hi(ra.z, hi[1]); hi[2] = mix_add_lo(ra.w, hi[2]); hi[3] = mix_add_hi(ra.w, hi[3]); \
                ra = from_qc(ld16(rr2, (evo_r2 | er.dm[PB]) + 64 * G, er.spix[PB] * ld22));                                                 \
            }                                                                                                       \
            st16(rout, (evo_out | er.dm[PB]) + 64 * G, er.spix[PB] * ldo2, to_qc(make_uint4(mix_pack_add(ra.x, lo[0], lo[1]), mix_pack_add(ra.y, lo[2], lo[3]),   \
                                                                     mix_pack_add(ra.z, hi[0], hi[1]), mix_pack_add(ra.w, hi[2], hi[3]))));  \
        } else {                                                                                                    \
        if constexpr (HAS_R1) {                                                                                     \
            const uint4 ra = from_qc(res[PB * NT + G]);                                                             \
            lo[0] += h16_lo(ra.x); lo[1] += h16_hi(ra.x); lo[2] += h16_lo(ra.y); lo[3] += h16_hi(ra.y);             \
            hi[0] += h16_lo(ra.z); hi[1] += h16_hi(ra.z); hi[2] += h16_lo(ra.w); hi[3] += h16_hi(ra.w);             \
            if constexpr (HAS_R2) {                                                                                 \
                const uint4 ca = from_qc(ld16(rr2, (evo_r2 | er.dm[PB]) + 64 * G, er.spix[PB] * ld22));                                     \
                lo[0] += h16_lo(ca.x); lo[1] += h16_hi(ca.x); lo[2] += h16_lo(ca.y); lo[3] += h16_hi(ca.y);         \
                hi[0] += h16_lo(ca.z); hi[1] += h16_hi(ca.z); hi[2] += h16_lo(ca.w); hi[3] += h16_hi(ca.w);         \
            }                                                                                                       \
        }                                                                                                           \
        st16(rout, (evo_out | er.dm[PB]) + 64 * G, er.spix[PB] * ldo2, to_qc(make_uint4(pk_h16(lo[0], lo[1]), pk_h16(lo[2], lo[3]), pk_h16(hi[0], hi[1]), pk_h16(hi[2], hi[3])))); \
        }                                                                                                           \
    }
            PP16_OUT(0, 0) PP16_OUT(0, 1) PP16_OUT(1, 0) PP16_OUT(1, 1) PP16_OUT(2, 0) PP16_OUT(2, 1) PP16_OUT(3, 0) PP16_OUT(3, 1)
#undef PP16_OUT
#undef PP16_GATHER
#undef SWAP16F
        } else {
            EpiRows er;
            epi_rows(cur, valid, er);
            // One 32x32 accumulator tile (tile row M, N-tile N) at a time, written out with static indices (a rolled loop
            // would push acc[][] to scratch).  COMPUTE layout after the first swap level: lane (r, h) holds, for
            // P = 0, 1, the 8 channels 16P + 8h .. +7 of pixel r (v0 / v1).  Residuals arrive in the memory layout
            // and are brought to the compute layout by the same (involutive) 16-lane swap that the packed results
            // take on their way out.
#define SWAP32(X, Y, LO, HI)                                                                                        \
    { const float xq = (X), yq = (Y);   /* scalars first: hipcc 7.2 mis-casts vector elements */                    \
      const v2u_t sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xq), __builtin_bit_cast(unsigned, yq), false, false); \
      const unsigned lo_ = sw.x, hi_ = sw.y; (LO) = __builtin_bit_cast(float, lo_); (HI) = __builtin_bit_cast(float, hi_); }
#define SWAP16(X, Y, A, B)                                                                                          \
    { const v2u_t sw = __builtin_amdgcn_permlane16_swap((X), (Y), false, false); (A) = sw.x; (B) = sw.y; }
#define PP_TILE(M, N)                                                                                               \
    if constexpr (N < NT) {                                                                                         \
        float v0[8], v1[8];                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                             \
            SWAP32(acc[M][N][j], acc[M][N][4 + j], v0[j], v0[4 + j])                                                \
            SWAP32(acc[M][N][8 + j], acc[M][N][12 + j], v1[j], v1[4 + j])                                           \
        }                                                                                                           \
        if constexpr (ACT && !POST) {                                                                               \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) { v0[j] = lrelu_f(v0[j], act_slope); v1[j] = lrelu_f(v1[j], act_slope); } \
        }                                                                                                           \
        unsigned p0[4], p1[4];       /* the packed results in the compute layout */                                 \
        if constexpr (HAS_R1 && RMIX) {                                                                             \
            const uint4 ra = from_qc(res[(M * NT + N) * 2]), rb = from_qc(res[(M * NT + N) * 2 + 1]);               \
            const unsigned ma[4] = {ra.x, ra.y, ra.z, ra.w}, mb[4] = {rb.x, rb.y, rb.z, rb.w};                      \
            unsigned xa[4], xb[4];                                                                                  \
            _Pragma("unroll") for (int d = 0; d < 4; ++d) SWAP16(ma[d], mb[d], xa[d], xb[d])                        \
            if constexpr (HAS_R2) {      /* (x + r1) in fp32 -- the bits of the two-step form --, then f16(. + r2) */  \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                     \
                    v0[2 * j] = mix_add_lo(xa[j], v0[2 * j]); v0[2 * j + 1] = mix_add_hi(xa[j], v0[2 * j + 1]);     \
                    v1[2 * j] = mix_add_lo(xb[j], v1[2 * j]); v1[2 * j + 1] = mix_add_hi(xb[j], v1[2 * j + 1]);     \
                }                                                                                                   \
                const uint4 ca = from_qc(ld16(rr2, (evo_r2 | er.dm[2 * M]) + 64 * N, er.spix[2 * M] * ld22)), cb = from_qc(ld16(rr2, (evo_r2 | er.dm[2 * M + 1]) + 64 * N, er.spix[2 * M + 1] * ld22));   \
                const unsigned na[4] = {ca.x, ca.y, ca.z, ca.w}, nb[4] = {cb.x, cb.y, cb.z, cb.w};                  \
                _Pragma("unroll") for (int d = 0; d < 4; ++d) SWAP16(na[d], nb[d], xa[d], xb[d])                    \
            }                                                                                                       \
            _Pragma("unroll") for (int d = 0; d < 4; ++d) {                                                         \
                p0[d] = mix_pack_add(xa[d], v0[2 * d], v0[2 * d + 1]);                                              \
                p1[d] = mix_pack_add(xb[d], v1[2 * d], v1[2 * d + 1]);                                              \
            }                                                                                                       \
        } else {                                                                                                    \
        if constexpr (HAS_R1) {                                                                                     \
            const uint4 ra = from_qc(res[(M * NT + N) * 2]), rb = from_qc(res[(M * NT + N) * 2 + 1]);               \
            const unsigned ma[4] = {ra.x, ra.y, ra.z, ra.w}, mb[4] = {rb.x, rb.y, rb.z, rb.w};                      \
            unsigned xa[4], xb[4];                                                                                  \
            _Pragma("unroll") for (int d = 0; d < 4; ++d) SWAP16(ma[d], mb[d], xa[d], xb[d])                        \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
                v0[2 * j] += h16_lo(xa[j]); v0[2 * j + 1] += h16_hi(xa[j]); \
                v1[2 * j] += h16_lo(xb[j]); v1[2 * j + 1] += h16_hi(xb[j]); \
            }                                                                                                       \
            if constexpr (HAS_R2) {                                                                                 \
                const uint4 ca = from_qc(ld16(rr2, (evo_r2 | er.dm[2 * M]) + 64 * N, er.spix[2 * M] * ld22)), cb = from_qc(ld16(rr2, (evo_r2 | er.dm[2 * M + 1]) + 64 * N, er.spix[2 * M + 1] * ld22));   \
                const unsigned na[4] = {ca.x, ca.y, ca.z, ca.w}, nb[4] = {cb.x, cb.y, cb.z, cb.w};                  \
                unsigned ya[4], yb[4];                                                                              \
                _Pragma("unroll") for (int d = 0; d < 4; ++d) SWAP16(na[d], nb[d], ya[d], yb[d])                    \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                     \
                    v0[2 * j] += h16_lo(ya[j]); v0[2 * j + 1] += h16_hi(ya[j]); \
                    v1[2 * j] += h16_lo(yb[j]); v1[2 * j + 1] += h16_hi(yb[j]); \
                }                                                                                                   \
            }                                                                                                       \
        }                                                                                                           \
        _Pragma("unroll") for (int d = 0; d < 4; ++d) { p0[d] = pk_h16(v0[2 * d], v0[2 * d + 1]); p1[d] = pk_h16(v1[2 * d], v1[2 * d + 1]); } \
        }                                                                                                           \
        unsigned s0[4], s1[4];                                                                                      \
        _Pragma("unroll") for (int d = 0; d < 4; ++d)                                                               \
            SWAP16(p0[d], p1[d], s0[d], s1[d])                                                                      \
        st16(rout, (evo_out | er.dm[2 * M]) + 64 * N, er.spix[2 * M] * ldo2, to_qc(make_uint4(s0[0], s0[1], s0[2], s0[3])));                     \
        st16(rout, (evo_out | er.dm[2 * M + 1]) + 64 * N, er.spix[2 * M + 1] * ldo2, to_qc(make_uint4(s1[0], s1[1], s1[2], s1[3])));                     \
    }
            if constexpr (POST && !POST_IN_M1) post_gemm();
            PP_TILE(0, 0) PP_TILE(0, 1) PP_TILE(1, 0) PP_TILE(1, 1)
#undef PP_TILE
#undef SWAP16
#undef SWAP32
        }
        store_chunk(pf0);
        cur = nxt;
        nxt = tile_origin(min(tile_of(k + 2), ntiles - 1));
        if constexpr (!PF0S1) load_chunk(nxt, 0, pf0);
        if constexpr (HAS_R1 && !(RES_IN_S1 && NCH == 2)) load_res(cur, tile_of(k + 1) < ntiles, 0, (PP_RES_SPLIT && NCH == 2) ? 1 : MTW);
        PP_BARRIER(3)
    }
#ifdef DSVC_PP_PROBE
    if (wg == 0 && lane == 0) {
        for (int i = 0; i < 4; ++i) { atomicAdd(&g_pp_phase[g * 8 + 2 * i], ppw[i]); atomicAdd(&g_pp_phase[g * 8 + 2 * i + 1], ppb[i]); }
        if (g == 0) atomicAdd(&g_pp_phase[16], (unsigned long long)iters);
    }
#endif

#ifdef DSVC_PP_CLOCK
    if (g == 0 && wg == 0 && lane == 0) {
        const unsigned long long ck_t1 = __builtin_amdgcn_s_memtime(), ck_r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&g_pp_clock[0], ck_t1 - ck_t0); atomicAdd(&g_pp_clock[1], ck_r1 - ck_r0); atomicAdd(&g_pp_clock[2], 1ull);
    }
#endif
    if (!g) lds_barrier();               // matches group 1's leading barrier
}

template <int NT, int NCH, bool HEAD, bool PRE, bool ACT, bool HAS_R1, bool HAS_R2, bool POST = false, bool M16 = false, bool QC = false, bool RMIX = false>
int launch_pp(hipStream_t st, const ConvParams& p, int ntiles, float act_slope, float pre_slope) {
#ifdef DSVC_PP_QC_VARIANT                       // measured slower (profiles/r05_pp_epilogue_experiments.txt): only in the variant build described in tools/bench_pp_opt.py
    if constexpr (!QC && !HEAD) {               // option "pp_qc": quad-contiguous epilogue (see the kernel's header comment)
        if (g_dsvc_opt.pp_qc) return launch_pp<NT, NCH, HEAD, PRE, ACT, HAS_R1, HAS_R2, POST, M16, true, RMIX>(st, p, ntiles, act_slope, pre_slope);
    }
#endif
    if constexpr (!RMIX && !HEAD && HAS_R1) {   // option "pp_mix": residual add + f16 packing as mixed-precision FMAs
        if (g_dsvc_opt.pp_mix) return launch_pp<NT, NCH, HEAD, PRE, ACT, HAS_R1, HAS_R2, POST, M16, QC, true>(st, p, ntiles, act_slope, pre_slope);
    }
    if constexpr (!M16 && !HEAD) {
        // option "pp16": 0 = 32x32x16 everywhere, 2 = 16x16x32 everywhere but the narrow heads, 1 (default) = where it measured faster
        // on the same box (tools/bench_pp16.py, 1088x1920, three interleaved rounds): one residual 187 -> 179 us, fused 1x1 tail
        // 212 -> 202 us; slower or equal elsewhere (plain 152 -> 153, pre + act 165 -> 174, two residuals 235 -> 254, 64 -> 32 99 -> 103)
        // [r5] re-checked on the round-5 kernels: isolated, the one-residual variant is now faster on 32x32x16 (172.8 vs 177.6 us, tools/bench_pp16.py),
        // but INSIDE the frame the 16x16x32 shape still wins (three alternating bench.py --lanes 1 runs per build on one box: dominant-conv
        // average 0.1775 vs 0.1787 ms, 27.9 vs 27.8 fps): the policy stays
        const bool faster16 = NCH == 2 && NT == 2 && !PRE && HAS_R1 && !HAS_R2;
        if (g_dsvc_opt.pp16 == 2 || (g_dsvc_opt.pp16 == 1 && faster16))
            return launch_pp<NT, NCH, HEAD, PRE, ACT, HAS_R1, HAS_R2, POST, true, QC, RMIX>(st, p, ntiles, act_slope, pre_slope);
    }
    const size_t lds = (size_t)NCH * 9 * NT * 32 * ROWB + (size_t)2 * PP_NPATCH * ROWB + (size_t)2 * NT * 32 * sizeof(float) +
                       (POST ? (size_t)NT * NT * 32 * ROWB : 0);
    static bool attr_set = false;
    auto kern = conv3x3_pp_kernel<NT, NCH, HEAD, PRE, ACT, HAS_R1, HAS_R2, POST, M16, QC, RMIX>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { dsvc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return DSVC_E_LAUNCH; }
        attr_set = true;
    }
    int grid = (ntiles + 1) / 2;
    static const int grid_cap = getenv("DSVC_PP_GRID") ? atoi(getenv("DSVC_PP_GRID")) : 256;
    if (grid > grid_cap) grid = grid_cap;                        // one persistent workgroup per CU, two tiles in flight each
    const int nband = grid % 8 == 0 ? 8 : 1, band_tiles = (ntiles + nband - 1) / nband;   // as in the kernel
    const int iters = (band_tiles + 2 * (grid / nband) - 1) / (2 * (grid / nband));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p, ntiles, iters, act_slope, pre_slope);
    DSVC_CHECK_LAUNCH("conv3x3_pp_kernel");
    return DSVC_OK;
}

template <int NT, bool PRE, bool ACT>
int launch_pp_res(hipStream_t st, const ConvParams& p, int ntiles, float a, float b) {
    if (p.res1 && p.res2) return launch_pp<NT, 2, false, PRE, ACT, true, true>(st, p, ntiles, a, b);
    if (p.res1) return launch_pp<NT, 2, false, PRE, ACT, true, false>(st, p, ntiles, a, b);
    return launch_pp<NT, 2, false, PRE, ACT, false, false>(st, p, ntiles, a, b);
}

template <int NT>
int launch_pp_act(hipStream_t st, const ConvParams& p, int ntiles, bool pre, bool act, float a, float b) {
    if (pre) return act ? launch_pp_res<NT, true, true>(st, p, ntiles, a, b) : launch_pp_res<NT, true, false>(st, p, ntiles, a, b);
    return act ? launch_pp_res<NT, false, true>(st, p, ntiles, a, b) : launch_pp_res<NT, false, false>(st, p, ntiles, a, b);
}

bool slope_of(int act, float slope, float* out) {
    if (act == DSVC_ACT_NONE) { *out = 1.f; return true; }       // max(v, v*1) == v
    if (act == DSVC_ACT_RELU) { *out = 0.f; return true; }
    if (act == DSVC_ACT_LRELU && slope >= 0.f && slope <= 1.f) { *out = slope; return true; }
    return false;
}

bool vec16_ok(const void* ptr, int ld) { return ((uintptr_t)ptr & 15) == 0 && (ld % 8) == 0; }

}  // namespace

// returns 1 when the shape is not covered (caller tries conv3x3_ws / the generic kernel), else DSVC_OK / error
int dsvc_conv3x3_pp_try(hipStream_t st, const ConvParams& p_in, const dsvc_conv_desc* d) {
    if (d->compute != DSVC_F16 || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->shuffle != 1) return 1;
    if (p_in.nchunks < 1 || p_in.nchunks > 2 || p_in.nseg > 2 || d->res1_scale) return 1;
    for (int s = 0; s < p_in.nseg; ++s) {                          // every pixel's chunk row is an aligned run of 16-B vectors
        const ConvSeg& sg = p_in.seg[s];
        if (sg.dtype != DSVC_F16 || (sg.C % 8) != 0 || !vec16_ok(sg.ptr, sg.ld)) return 1;
        if (p_in.nchunks == 2 && (sg.C % 32) != 0) return 1;
    }
    ConvParams p = p_in;
    p.tiles_x = (p.Wo + 31) / 32;
    p.tiles_y = (p.Ho + PP_TH - 1) / PP_TH;
    const int ntiles = p.B * p.tiles_x * p.tiles_y;
    if ((int64_t)p.B * p.H * p.W >= (1 << 24)) return 1;   // pixel indices go through the 24-bit multiplier
    if (ntiles < 1024) return 1;           // small images: the generic kernel's finer tiling fills the chip better
    if (d->post_weight) {                  // fused ResBottleneck tail: conv3x3 -> act -> conv1x1 -> + residual
        float a_slope;
        if (p.nchunks != 2 || d->Cout != 64 || d->pre_act != DSVC_ACT_NONE || d->res2 || !slope_of(d->act, d->slope, &a_slope)) return 1;
        if (d->out_dtype != DSVC_F16 || !vec16_ok(d->out, d->out_ld) || ((uintptr_t)d->post_weight & 15) != 0) return 1;
        if (d->res1 && (d->res1_dtype != DSVC_F16 || !vec16_ok(d->res1, d->res1_ld))) return 1;
        const bool act1 = d->act != DSVC_ACT_NONE;
        if (d->res1) return act1 ? launch_pp<2, 2, false, false, true, true, false, true>(st, p, ntiles, a_slope, 1.f)
                                 : launch_pp<2, 2, false, false, false, true, false, true>(st, p, ntiles, 1.f, 1.f);
        return act1 ? launch_pp<2, 2, false, false, true, false, false, true>(st, p, ntiles, a_slope, 1.f)
                    : launch_pp<2, 2, false, false, false, false, false, true>(st, p, ntiles, 1.f, 1.f);
    }
    if (d->Cout <= 8) {                    // narrow heads (flow / frame / mask outputs): any activation, element-wise stores
        if (d->pre_act != DSVC_ACT_NONE || d->res2) return 1;
        return p.nchunks == 2 ? launch_pp<1, 2, true, false, false, false, false>(st, p, ntiles, 1.f, 1.f)
                              : launch_pp<1, 1, true, false, false, false, false>(st, p, ntiles, 1.f, 1.f);
    }
    if (d->Cout != 32 && d->Cout != 64) return 1;
    if (d->out_dtype != DSVC_F16 || !vec16_ok(d->out, d->out_ld)) return 1;
    if (d->res1 && (d->res1_dtype != DSVC_F16 || !vec16_ok(d->res1, d->res1_ld))) return 1;
    if (d->res2 && (!d->res1 || d->res2_dtype != DSVC_F16 || !vec16_ok(d->res2, d->res2_ld))) return 1;
    float act_slope, pre_slope;
    if (!slope_of(d->act, d->slope, &act_slope) || !slope_of(d->pre_act, d->slope, &pre_slope)) return 1;
    const bool pre = d->pre_act != DSVC_ACT_NONE, act = d->act != DSVC_ACT_NONE;
    if (p.nchunks == 1) {                  // first-layer convs (8 / 16 / 32 input channels): no fused pre-activation or residual
        if (pre || d->res1) return 1;
        if (d->Cout == 64) return act ? launch_pp<2, 1, false, false, true, false, false>(st, p, ntiles, act_slope, 1.f)
                                      : launch_pp<2, 1, false, false, false, false, false>(st, p, ntiles, 1.f, 1.f);
        return act ? launch_pp<1, 1, false, false, true, false, false>(st, p, ntiles, act_slope, 1.f)
                   : launch_pp<1, 1, false, false, false, false, false>(st, p, ntiles, 1.f, 1.f);
    }
    if (d->Cout == 64) return launch_pp_act<2>(st, p, ntiles, pre, act, act_slope, pre_slope);
    return launch_pp_act<1>(st, p, ntiles, pre, act, act_slope, pre_slope);
}
