// kernels_deep.h -- D time steps per pass, D = 6, 7, 8, ...: the marching kernel of kernels_step5.h / kernels_step6.h written once
// for any depth (round 5).  Read those headers first: overlapping strips (a wave's 64 lanes x 4 cells are the strip and its skirt),
// segment pairs (the two waves of a workgroup start back to back at the pair's middle line and hand each other the links that
// cross it while their pipelines fill), peeled pipeline fill, stage windows.
//
// What round 5 measured (profiles/r05_experiments.txt):
//   * k_step6 is bound by INSTRUCTION ISSUE, not by HBM: with every global access removed (diagnostic build) its launch still takes
//     0.86 ms of 1.08; with the arithmetic removed 0.92.  At six waves per CU two SIMDs of every CU hold two waves and two hold one,
//     all waves have the same rows to march, and the pairs set the pace.
//   * A gfx950 SIMD issues one vector instruction per ~2.5 cycles (v_pk_*_f32: ~3.5) when TWO waves feed it, and a single wave no
//     faster than one per ~5 cycles -- any instruction, packed or not, vector, scalar or LDS (tools/valu_issue_probe.hip).
//   So the kernel is parametrised on where a wave's state lives:
//     RW   stage windows kept in registers (the first RW of the D - 1), the others in wave-private LDS: eight 1-KiB slots each
//          (links 1, 3 of the newest row and the two-row ring of the three links pulled from behind; link 0, which no stage shifts,
//          stays in four registers per window) -- five of them fill the 40 KB a wave has at four waves per CU: D = 7 with RW = 1;
//     PFD  rows gathered ahead (0: none -- two waves per SIMD cover each other's waits; 1: the next row's gather is in flight while
//          this one is computed -- ONE wave per SIMD, 512 registers, 40 KB of LDS: __launch_bounds__(128, 1)).
//   The launch moves the same 72 B per cell whatever D is.  Same cell functions as every kernel: bitwise equal to k_step.
//
// Stage S (2..D) of an iteration reads window S - 1: links 0,1,3 of the row it is about to advance (d), the three links pulled from
// behind of the row before that (g), and takes the three links pulled from ahead out of the row stage S - 1 has just produced; that
// row then enters the window.  The skirt is D - 1 cells deep: two lanes (8 cells) for D <= 9, strips 240 cells apart (STEP6_VALID).
#pragma once

#ifdef LB_DIAG
#define LB_DEEP_NOCOLLIDE if (!(a.diag & 1))
#else
#define LB_DEEP_NOCOLLIDE
#endif

namespace {

// The skirt is D - 1 cells deep, i.e. whole lanes of four cells that are computed and never stored, at either end of a strip
#ifdef LB_DEEP_SKIRT_LANES                               // (experiment: more skirt than needed, e.g. 4 lanes = strips 224 cells = 7 x 128 B apart)
constexpr int deep_skirt_lanes(int D) { return LB_DEEP_SKIRT_LANES; }
#else
constexpr int deep_skirt_lanes(int D) { return (D - 1 + 3) / 4; }
#endif
constexpr int deep_valid(int D) { return STRIP_W - 8 * deep_skirt_lanes(D); }           // cells stored per strip and row (D = 6..9: 240)
constexpr int deep_strips(int nx, int D) { return (nx + deep_valid(D) - 1) / deep_valid(D); }
// Where a wave's state lives, per depth (see the header comment): windows in registers, rows gathered ahead
#ifndef LB_DEEP_RW
#define LB_DEEP_RW 1                    // (2: the form of round 5's first D = 7 kernel, for tools/r06/rw2_check.sh)
#endif
constexpr int deep_rw(int D) { return LB_DEEP_RW; }
constexpr int DEEP_WSLOTS = 8;          // LDS slots of a stage window
constexpr int deep_pfd(int D) { return 1; }
// Code footprint (two CUs share a 64 KB instruction cache; a lone wave has nobody to cover its fetch misses; 8192^2, k MLUPS,
// profiles/r05_footprint.txt): the boundary rule out of line (boundary_rule_call): pipe 306 -> 383-393, cavity 319 -> 393-403; the
// steady iterations in PAIRS (the two row buffers swap roles, the LDS ring slots become immediate offsets: -1.3 % instructions, twice
// the loop) pay without an obstacle mask (periodic 435 against 412, cavity 403 / 383) and cost with one (periodic + mask 350 against
// 368, pipe + mask 4096^2 250 / 263); a code path of their own for the strips without a wall column (no per-lane wall test: -36
// instructions per row) costs more in footprint than it saves (pipe 337 against 393 with pairs, 388 / 384 without): off.
// Since the row in flight sits in accumulation registers (LB_DEEP_MANUAL below) no buffers swap roles any more, the pairs only
// make the ring slots immediate: periodic 458 against 452 k, but the walled kernels (2,600 instructions per iteration: the pair of
// pairs of the two waves is ~62 KB of code) now lose by them: pipe 8192^2 392 against 384, cavity 418 / 406, pipe 4096^2 337 / 329
// (profiles/r05_pairs_ab.txt).  Mode 1: pairs in periodic boxes without a mask only (2: wherever there is no mask; 0: nowhere).
#ifndef LB_DEEP_PAIRS_MODE
#define LB_DEEP_PAIRS_MODE 1
#endif
constexpr bool deep_pairs(bool mask, int bc) { return LB_DEEP_PAIRS_MODE == 2 ? !mask : (LB_DEEP_PAIRS_MODE == 1 ? !mask && bc == LB_BC_PERIODIC : false); }

template <int RW, int NL>
struct DeepState {
    Window w[RW > 0 ? RW : 1];          // stage windows 1..RW (registers)
    f4a d0[NL];                         // link 0 of the newest row of the LDS windows RW+1..D-1
    unsigned mhist;                     // obstacle-mask history (per byte: bit j = the row loaded j iterations ago, j = 1..D-1)
};
typedef unsigned u4v __attribute__((ext_vector_type(4)));
struct DeepCtx {
    int lane, x4, ym, n_iter;
    bool store_lane;
    f4a (*mine)[64], (*other)[64];      // my LDS windows RW+1..D-1 (DEEP_WSLOTS slots each, in that order), the other wave's
    f4a (*ho)[64];                      // k_deep2: the slot a front wave hands its rows to its back wave through (9 links + the obstacle flags)
    f4a (*dma)[64];                     // k_deep2, front waves: the slot the row gathered ahead is loaded INTO (buffer_load ... lds), and
    unsigned dma_off;                   //   its byte offset inside the workgroup's LDS (what M0 takes)
};
// k_deep2 (below): a strip's march split between two waves -- the front wave runs stages 1..F, the back wave stages F+1..D
constexpr int DEEP_WHOLE = 0, DEEP_FRONT = 1, DEEP_BACK = 2;
constexpr int DEEP_HO_SLOTS = 10;

// An LDS window: slot 0, 1 = links 1, 3 of the newest row; slots 2..4 / 5..7 = links A, B, C (pulled from behind) of the newest two
// rows, a ring: the iteration's parity picks the OLDER row, which is read, then overwritten
__device__ __forceinline__ void deep_window_load(f4a (*W)[64], int lane, int it, f4a d0, Window &w)
{
    const int gs = 2 + 3 * (it & 1);
    w.d0 = d0; w.d1 = W[0][lane]; w.d3 = W[1][lane];
    w.g2 = W[gs][lane]; w.g5 = W[gs + 1][lane]; w.g6 = W[gs + 2][lane];
}
// (Round 5 also read the window rows SHIFTED -- a slot is a row of 256 cells, the cell to the left of each of a lane's cells is the
//  same row read 4 bytes lower: gfx950 serves a ds_read_b128 at any dword address, tools/lds_unaligned_probe.hip -- through asm, one
//  stage ahead behind scheduling barriers: 100 instructions per row fewer (three moves and a DPP per shifted link) and 5 % SLOWER,
//  k_deep<7> 8192^2 407 against 430 k MLUPS on one box: the unaligned reads cost more LDS passes than the moves cost issue slots.
//  profiles/r05_shifted_reads_ab.txt; not kept.)
template <bool DOWN>
__device__ __forceinline__ void deep_window_push(f4a (*W)[64], int lane, int it, f4a &d0, const f4a (&q)[9])
{
    typedef Dir<DOWN> D_;
    const int gs = 2 + 3 * (it & 1);
    d0 = q[0]; W[0][lane] = q[1]; W[1][lane] = q[3];
    W[gs][lane] = q[D_::A]; W[gs + 1][lane] = q[D_::B]; W[gs + 2][lane] = q[D_::C];
}
// the three links of row q that cross the pair's middle line, into the OTHER wave's window at ring slots gs..gs+2
template <bool DOWN>
__device__ __forceinline__ void deep_publish(f4a (*W)[64], int lane, int gs, const f4a (&q)[9])
{
    typedef Dir<DOWN> D_;
    W[gs][lane] = q[D_::An]; W[gs + 1][lane] = q[D_::Bn]; W[gs + 2][lane] = q[D_::Cn];
}

// ---- a row's nine gathers and nine stores through BUFFER instructions (round 5) -----------------------------------------------
// A global access takes a 64-bit base per instruction: nine bases per gathered row and nine lane offsets per stored row, ~55 scalar and
// ~33 vector instructions of address arithmetic per row -- to a wave that pays ~5 cycles for each.  A buffer access adds three things
// itself: a resource (its base: a source row's start moved one float down, so that the displaced planes need no negative offset;
// three per gathered row, one per stored row), a scalar offset per plane (eight loop-invariant scalars) and a small immediate
// (0 / 4 / 8 bytes: pulled from the left / same column / from the right); the lane offset is one register for all.
#ifndef LB_DEEP_BUFFER
#define LB_DEEP_BUFFER 1
#endif
// Extent of every resource: the whole 32-bit range.  A raw buffer access is range-checked as offset >= num_records - soffset, and the
// scalar offset carries the plane: up to 8 planes x 4 bytes, which in the planar layout (LB_FLAG_PLANAR) of an 8192^2 lattice is 2.16 GB
// -- with 2 GiB of records (round 5) plane 8 of such a lattice would have read zeros and dropped its stores.  The host admits a
// lattice to these kernels only while (8 planes + a row) x 4 bytes < 4 GiB (marching_planes_fit, lb_hip.cpp).
constexpr int DEEP_NUM_RECORDS = -1;                     // 0xffffffff bytes
// (a resource must live in scalar registers; where the compiler cannot see that a row's base is wave-uniform it wraps every access in
//  a "waterfall" loop over the distinct values -- 15 of them per row pair in the first form of this code: say so explicitly)
__device__ __forceinline__ float *deep_uniform(const float *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<float *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ f4a deep_buf_load(__amdgpu_buffer_rsrc_t r, int vo, int so, int imm)
{
    return __builtin_bit_cast(f4a, __builtin_amdgcn_raw_buffer_load_b128(r, vo + imm, so, 0));
}
// row1_load (kernels_step4.h) for the strips of k_deep: no halo cell; the nine plane loads through one buffer resource
template <int BC, bool MASK>
__device__ __forceinline__ void deep_row_load(const StepArgs &a, int r, int x4, Row1 &o)
{
    int ym, yp;
    o.have = step1_rows(a, r, o.rr, ym, yp);
    o.mk = uc4{0, 0, 0, 0};
    o.hsolid = false;
    o.hxc = -1;
    if (o.have) {
        if (LB_DEEP_BUFFER) {
            const long long P = a.pitch, S = a.plane;
            const float *s = a.src;
            const int yl = o.rr;
            o.wp = WrapPatch{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (BC == LB_BC_PERIODIC) {                     // (the seam lanes' wrap elements: as gather_issue)
                const int c = a.nx - 1 - x4;
                const bool wrap_w = x4 == 0, wrap_e = c >= 0 && c < 4;
                if (wrap_w) {
                    o.wp.p1 = s[1 * S + (long long)yl * P + a.nx - 1];
                    o.wp.p5 = s[5 * S + (long long)ym * P + a.nx - 1];
                    o.wp.p8 = s[8 * S + (long long)yp * P + a.nx - 1];
                }
                if (wrap_e) {
                    o.wp.w3 = s[3 * S + (long long)yl * P];
                    o.wp.w6 = s[6 * S + (long long)ym * P];
                    o.wp.w7 = s[7 * S + (long long)yp * P];
                }
            }
            // one resource per source row (the row itself, the rows its cy = +1 / cy = -1 links come from: wrapped by step1_rows where
            // the box is periodic), each based one float BELOW the row start: the immediate is 0 / 4 / 8 for a pull from the left / the
            // same column / the right; the scalar offset is the plane's
            const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(deep_uniform(s + (long long)yl * P - 1), 0, DEEP_NUM_RECORDS, 0x00020000);
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(deep_uniform(s + (long long)ym * P - 1), 0, DEEP_NUM_RECORDS, 0x00020000);
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(deep_uniform(s + (long long)yp * P - 1), 0, DEEP_NUM_RECORDS, 0x00020000);
            const unsigned S4 = (unsigned)a.plane * 4u;
            const int vo = x4 * 4;
            o.q[0] = deep_buf_load(r0, vo, 0, 4);
            o.q[1] = deep_buf_load(r0, vo, (int)(S4), 0);
            o.q[2] = deep_buf_load(rm, vo, (int)(2u * S4), 4);
            o.q[3] = deep_buf_load(r0, vo, (int)(3u * S4), 8);
            o.q[4] = deep_buf_load(rp, vo, (int)(4u * S4), 4);
            o.q[5] = deep_buf_load(rm, vo, (int)(5u * S4), 0);
            o.q[6] = deep_buf_load(rm, vo, (int)(6u * S4), 8);
            o.q[7] = deep_buf_load(rp, vo, (int)(7u * S4), 8);
            o.q[8] = deep_buf_load(rp, vo, (int)(8u * S4), 0);
            if (MASK) o.mk = *reinterpret_cast<const uc4 *>(lane_ptr(a.mask + (long long)yl * a.fpitch, x4));
        } else {
            gather_issue<BC, MASK, false>(a, x4, o.rr, ym, yp, o.q, o.mk, o.wp);
        }
    } else {
        o.wp = WrapPatch{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 9; ++k) o.q[k] = f4a{0.f, 0.f, 0.f, 0.f};
    }
}
// ---- the row gathered AHEAD, waited for by hand (round 5) --------------------------------------------------------------------------
// The compiler's wait-count pass counts a load's younger operations exactly inside one basic block only: a row gathered in one
// iteration and consumed in the next is waited for with a count that forgets the nine stores (and, issued first, the nine new loads)
// between the two -- `s_waitcnt vmcnt(0..8)` where 9..18 would do: the wave drains what it has just issued and the gather "ahead"
// hides nothing (SQ_WAIT_ANY 22 % of the wave cycles of k_deep<7>, profiles/r05_sq_deep7_8192.txt; a two-buffer loop of 30 lines
// shows the same counts).  So the row in flight is hidden from the compiler altogether: its loads are issued by an asm block into a
// FIXED window of accumulation registers, a[0:42] (rounds 5: a[192:234] -- which made every wave ALLOCATE 235 accumulation registers
// beside its ~230 vector registers: a SIMD's whole file, so that not even the 256-thread copy kernels of a halo exchange found a
// place beside a marching wave; at the bottom of the file a wave takes ~280 of the 512, round 6), that the kernel uses for nothing else
// -- it has no accumulation-register spills; the day it has, they start at a0 and the check below refuses the build -- (tools/check_agpr_window.py checks the
// disassembly for strays), and taken out of it behind an `s_waitcnt vmcnt(N)` written here, N = the vector-memory instructions this
// wave has issued since the row's last load -- the nine (MACRO: twelve) stores of the iteration in between, every steady iteration
// issues them; none in the filling iterations --: vector-memory operations of a wave complete in issue order (loads and stores count
// together), "all but the N youngest are done" is then exactly "the row has arrived".  A count too SMALL only waits longer; one too
// LARGE would read registers still in flight, hence: N is 0 wherever the iteration before was not a steady one (deep_march drains the
// counter once before its loop), and LB_DIAG builds, whose ablation bits skip loads and stores, do not use this path.
#ifndef LB_DEEP_MANUAL
#ifdef LB_DIAG
#define LB_DEEP_MANUAL 0
#else
#define LB_DEEP_MANUAL LB_DEEP_BUFFER
#endif
#endif
__device__ __forceinline__ u4v deep_rsrc_words(const float *p)
{
    const unsigned long long v = (unsigned long long)p;
    return u4v{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)v), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)), (unsigned)DEEP_NUM_RECORDS,
               0x00020000u};
}
// issues row r's loads into a[0:35] (planes 0..8), a[36:41] (the seam lanes' wrap elements), a42 (obstacle flags); o: rr, have
template <int BC, bool MASK>
__device__ __forceinline__ void deep_row_issue(const StepArgs &a, int r, int x4, Row1 &o)
{
    int ym, yp;
    o.have = step1_rows(a, r, o.rr, ym, yp);
    o.hsolid = false;
    o.hxc = -1;
    if (o.have) {
        const long long P = a.pitch, S = a.plane;
        const float *s = a.src;
        const int yl = o.rr;
        if (BC == LB_BC_PERIODIC) {                     // (issued first, as gather_issue does)
            const int c = a.nx - 1 - x4;
            const bool wrap_w = x4 == 0, wrap_e = c >= 0 && c < 4;
            if (wrap_w) {
                const float *e = s + a.nx - 1;
                asm volatile("global_load_dword a36, %0, off\n\tglobal_load_dword a37, %1, off\n\tglobal_load_dword a38, %2, off"
                             :: "v"(e + 1 * S + (long long)yl * P), "v"(e + 5 * S + (long long)ym * P), "v"(e + 8 * S + (long long)yp * P)
                             : "memory", "a36", "a37", "a38");
            }
            if (wrap_e) {
                asm volatile("global_load_dword a39, %0, off\n\tglobal_load_dword a40, %1, off\n\tglobal_load_dword a41, %2, off"
                             :: "v"(s + 3 * S + (long long)yl * P), "v"(s + 6 * S + (long long)ym * P), "v"(s + 7 * S + (long long)yp * P)
                             : "memory", "a39", "a40", "a41");
            }
        }
        const u4v r0 = deep_rsrc_words(s + (long long)yl * P - 1), rm = deep_rsrc_words(s + (long long)ym * P - 1),
                  rp = deep_rsrc_words(s + (long long)yp * P - 1);
        const unsigned S4 = (unsigned)a.plane * 4u;
        asm volatile("buffer_load_dwordx4 a[0:3], %0, %1, 0 offen offset:4\n\t"
                     "buffer_load_dwordx4 a[4:7], %0, %1, %4 offen\n\t"
                     "buffer_load_dwordx4 a[8:11], %0, %2, %5 offen offset:4\n\t"
                     "buffer_load_dwordx4 a[12:15], %0, %1, %6 offen offset:8\n\t"
                     "buffer_load_dwordx4 a[16:19], %0, %3, %7 offen offset:4\n\t"
                     "buffer_load_dwordx4 a[20:23], %0, %2, %8 offen\n\t"
                     "buffer_load_dwordx4 a[24:27], %0, %2, %9 offen offset:8\n\t"
                     "buffer_load_dwordx4 a[28:31], %0, %3, %10 offen offset:8\n\t"
                     "buffer_load_dwordx4 a[32:35], %0, %3, %11 offen"
                     :: "v"(x4 * 4), "s"(r0), "s"(rm), "s"(rp), "s"(S4), "s"(2u * S4), "s"(3u * S4), "s"(4u * S4), "s"(5u * S4), "s"(6u * S4),
                        "s"(7u * S4), "s"(8u * S4)
                     : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13",
                       "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28",
                       "a29", "a30", "a31", "a32", "a33", "a34", "a35");
        if (MASK) asm volatile("global_load_dword a42, %0, off" :: "v"(lane_ptr(a.mask + (long long)yl * a.fpitch, x4)) : "memory", "a42");
    }
}
// the row issued last: waits until all but the wave's N youngest vector-memory operations are done, then takes it out of the window
template <int BC, bool MASK, int N>
__device__ __forceinline__ void deep_row_take(Row1 &o)
{
    static_assert(N == 0 || N == 9 || N == 12, "the stores of one steady iteration, or nothing");
#define LB_DEEP_TAKE_OUTS "={a[0:3]}"(o.q[0]), "={a[4:7]}"(o.q[1]), "={a[8:11]}"(o.q[2]), "={a[12:15]}"(o.q[3]), \
                          "={a[16:19]}"(o.q[4]), "={a[20:23]}"(o.q[5]), "={a[24:27]}"(o.q[6]), "={a[28:31]}"(o.q[7]), "={a[32:35]}"(o.q[8])
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" : LB_DEEP_TAKE_OUTS :: "memory");
#ifdef LB_DEEP_TIMING_NO_WAIT                        // timing only, wrong results: what the wait itself costs
    else if (N == 9) asm volatile("s_waitcnt vmcnt(63)" : LB_DEEP_TAKE_OUTS :: "memory");
#endif
    else if (N == 9) asm volatile("s_waitcnt vmcnt(9)" : LB_DEEP_TAKE_OUTS :: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" : LB_DEEP_TAKE_OUTS :: "memory");
#undef LB_DEEP_TAKE_OUTS
    o.wp = WrapPatch{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (BC == LB_BC_PERIODIC)
        asm volatile("" : "={a36}"(o.wp.p1), "={a37}"(o.wp.p5), "={a38}"(o.wp.p8), "={a39}"(o.wp.w3), "={a40}"(o.wp.w6), "={a41}"(o.wp.w7));
    o.mk = uc4{0, 0, 0, 0};
    if (MASK) {
        unsigned m;
        asm volatile("" : "={a42}"(m));
        o.mk = __builtin_bit_cast(uc4, m);
    }
    if (!o.have) {                                      // (a row outside a walled box: nothing was issued)
#pragma unroll
        for (int k = 0; k < 9; ++k) o.q[k] = f4a{0.f, 0.f, 0.f, 0.f};
        o.mk = uc4{0, 0, 0, 0};
    }
}

// ---- the row gathered ahead, straight into LDS (k_deep2's front waves; round 6) ---------------------------------------------------
// A front wave shares its SIMD with another wave: 256 registers.  The accumulation-register window above does not survive that -- a
// kernel that names accumulation registers at two waves per SIMD is given 128 vector + 128 accumulation registers (LLVM's default
// split; "amdgpu-agpr-alloc"="44" mends it to 212 + 44, but then the compiler's own spills land in a0..a43, i.e. in the window) --,
// and the same window in vector registers does not either (the compiler copies the values "in" it around branches: data that has
// not arrived yet).  So the row in flight is in no register at all: `buffer_load_dwordx4 ... lds` writes a lane's 16 bytes to LDS at
// M0 + 16 x lane -- measured, tools/lds_dma_probe.hip / profiles/r06_lds_dma_probe.txt: M0 may lie beyond 64 KB; an instruction offset
// moves the LDS address as well as the global one (hence none here: the 0 / 4 / 8-byte displacement of the pulled planes sits in three
// lane-offset registers); lanes masked out of exec write nothing --, one 1-KiB slot per plane, a tenth for the row's obstacle flags and
// the seam lanes' wrap elements.  The loads are issued by asm blocks (the compiler must not count them: kernels_deep.h above), taken
// behind `s_waitcnt vmcnt(0)` -- a front wave has nothing else in flight: it stores nothing -- by nine ds_read_b128 where the
// accumulation registers took 42 v_accvgpr_read.
template <int BC, bool MASK>
__device__ __forceinline__ void deep_row_issue_lds(const StepArgs &a, int r, int x4, unsigned slot, Row1 &o)
{
    int ym, yp;
    o.have = step1_rows(a, r, o.rr, ym, yp);
    o.hsolid = false;
    o.hxc = -1;
    if (o.have) {
        const long long P = a.pitch, S = a.plane;
        const float *s = a.src;
        const int yl = o.rr;
        unsigned m0_kept;                               // (M0 is the compiler's: every block puts back what it found there)
        // (whatever still reads the slot -- the previous row's take -- has its data first)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (BC == LB_BC_PERIODIC) {
            const int c = a.nx - 1 - x4;
            const bool wrap_w = x4 == 0, wrap_e = c >= 0 && c < 4;
            if (wrap_w) {
                const float *e = s + a.nx - 1;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\t"
                             "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\t"
                             "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %3, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(m0_kept)
                             : "v"(e + 1 * S + (long long)yl * P), "v"(e + 5 * S + (long long)ym * P), "v"(e + 8 * S + (long long)yp * P),
                               "s"(slot + 9 * 1024 + 256)
                             : "memory", "scc");
            }
            if (wrap_e) {
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\t"
                             "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\t"
                             "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %3, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(m0_kept)
                             : "v"(s + 3 * S + (long long)yl * P), "v"(s + 6 * S + (long long)ym * P), "v"(s + 7 * S + (long long)yp * P),
                               "s"(slot + 9 * 1024 + 256)
                             : "memory", "scc");
            }
        }
        const u4v r0 = deep_rsrc_words(s + (long long)yl * P - 1), rm = deep_rsrc_words(s + (long long)ym * P - 1),
                  rp = deep_rsrc_words(s + (long long)yp * P - 1);
        const unsigned S4 = (unsigned)a.plane * 4u;
        const int v0 = x4 * 4, v4 = v0 + 4, v8 = v0 + 8;       // pulled from the left / the same column / from the right
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %15\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, 0 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %4, %7 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %5, %8 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %4, %9 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %6, %10 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %5, %11 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %12 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %6, %13 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %6, %14 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_kept)
                     : "v"(v0), "v"(v4), "v"(v8), "s"(r0), "s"(rm), "s"(rp), "s"(S4), "s"(2u * S4), "s"(3u * S4), "s"(4u * S4), "s"(5u * S4),
                       "s"(6u * S4), "s"(7u * S4), "s"(8u * S4), "s"(slot)
                     : "memory", "scc");
        if (MASK)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_kept) : "v"(lane_ptr(a.mask + (long long)yl * a.fpitch, x4)), "s"(slot + 9 * 1024) : "memory");
    }
}
// the row issued last: everything this wave has in flight is that row
template <int BC, bool MASK>
__device__ __forceinline__ void deep_row_take_lds(const f4a (*S)[64], int lane, Row1 &o)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (o.have) {
#pragma unroll
        for (int k = 0; k < 9; ++k) o.q[k] = S[k][lane];
        const unsigned *X = reinterpret_cast<const unsigned *>(S[9]);
        o.mk = MASK ? __builtin_bit_cast(uc4, X[lane]) : uc4{0, 0, 0, 0};
        o.wp = WrapPatch{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (BC == LB_BC_PERIODIC) {                     // (the lane at x = 0 and the lane at x = nx - 4 share the three sub-slots)
            const float e1 = __builtin_bit_cast(float, X[64 + lane]), e2 = __builtin_bit_cast(float, X[128 + lane]),
                        e3 = __builtin_bit_cast(float, X[192 + lane]);
            o.wp = WrapPatch{e1, e2, e3, e1, e2, e3};
        }
    } else {                                            // (a row outside a walled box: nothing was issued)
        o.wp = WrapPatch{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 9; ++k) o.q[k] = f4a{0.f, 0.f, 0.f, 0.f};
        o.mk = uc4{0, 0, 0, 0};
    }
}

// store_row9 (kernels_fused.h) through a buffer resource based at the row
__device__ __forceinline__ void deep_row_store(const StepArgs &a, int r, int x4, const f4a (&t)[9])
{
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(deep_uniform(a.dst + (long long)r * a.pitch), 0, DEEP_NUM_RECORDS, 0x00020000);
    const unsigned S4 = (unsigned)a.plane * 4u;
    const int vo = x4 * 4;
    // (a compiler-level memory clobber, as store_row9's: without one in the loop the optimiser promotes the wave-private LDS windows
    //  -- written in one iteration, read two later, clobbered by nothing it can see -- into "registers": 256 + 256 of them and 1.4 KB
    //  of scratch per lane)
    asm volatile("" ::: "memory");
    if (a.nts != 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, t[k]), rd, vo, (int)(k * S4), 2);     // (2: nt)
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, t[k]), rd, vo, (int)(k * S4), 0);
    }
}

// Stages S..D of one iteration, S >= 2.  qin = the row stage S - 1 produced in this iteration (position i - (S - 2)).
template <int BC, bool MASK, bool MACRO, int D, int RW, bool DOWN, int NST, int S, int ROLE = DEEP_WHOLE>
__device__ __forceinline__ void deep_stage(const StepArgs &a, const DeepCtx &cx, const int i, const int it,
                                           DeepState<RW, D - 1 - RW> &st, f4a (&qin)[9], f4a &r4, f4a &u4, f4a &v4)
{
    if constexpr (S <= D) {
        constexpr int K = S - 1;                                   // the window this stage reads
        const int lane = cx.lane, x4 = cx.x4;
        constexpr int L = K > RW ? K - RW - 1 : 0;                 // my LDS window's index (meaningful for K > RW only)
        f4a (*W)[64] = cx.mine + L * DEEP_WSLOTS;
        if constexpr (NST >= S) {
            int r, t0_, t1_;
            (void)step1_rows(a, DOWN ? cx.ym - 1 - (i - K) : cx.ym + (i - K), r, t0_, t1_);
            f4a t[9];
            // (every window takes its new row as soon as its old one has been gathered from: qin dies here)
            if constexpr (K <= RW) {
                skirt_gather<DOWN>(st.w[K - 1], qin, t);
                window_push_dir<DOWN>(st.w[K - 1], qin);
            } else {
                Window w;
                deep_window_load(W, lane, it, st.d0[L], w);
                skirt_gather<DOWN>(w, qin, t);
                deep_window_push<DOWN>(W, lane, it, st.d0[L], qin);
            }
            LB_DEEP_NOCOLLIDE collide_row<BC, MASK, true>(a, x4, a.y0 + r, t, mask_bits(st.mhist, K), r4, u4, v4);
            if constexpr (S == D && ROLE == DEEP_FRONT) {
                // k_deep2: the row goes to my back wave (every lane's: its skirt shifts need the skirt lanes' cells), with its obstacle flags
#pragma unroll
                for (int k = 0; k < 9; ++k) cx.ho[k][lane] = t[k];
                if (MASK) reinterpret_cast<unsigned *>(cx.ho[9])[lane] = __builtin_bit_cast(unsigned, mask_bits(st.mhist, K));
            } else if constexpr (S == D) {
#ifdef LB_DIAG
                if (!(a.diag & (1 << 22)))
#endif
                if (cx.store_lane) {
                    float *d = a.dst + (long long)r * a.pitch;     // row start, uniform
                    if (LB_DEEP_BUFFER) deep_row_store(a, r, x4, t);
                    else store_row9<false>(a.nts != 0, d, a.plane, x4, t);
                    if (MACRO) {
                        const long long m = (long long)r * a.fpitch;
                        store4<false>(lane_ptr(a.rho + m, x4), r4);
                        store4<false>(lane_ptr(a.u + m, x4), u4);
                        store4<false>(lane_ptr(a.v + m, x4), v4);
                    }
                }
            } else {
                if constexpr (NST == S) {
                    // my position 0 after step S -> what the other wave's stage S + 1 pulls from behind ITS position 0: into its
                    // window S -- the ring row it reads as "the older one" in its next iteration (i = S: 2 + 3 (S & 1)) --, or,
                    // a register window, through its still idle last LDS window (read at the top of its next iteration)
                    if constexpr (S <= RW) deep_publish<DOWN>(cx.other + (D - 2 - RW) * DEEP_WSLOTS, lane, S == 1 ? 5 : 2, t);
                    else deep_publish<DOWN>(cx.other + (S - RW - 1) * DEEP_WSLOTS, lane, 2 + 3 * (S & 1), t);
                }
                deep_stage<BC, MASK, MACRO, D, RW, DOWN, NST, S + 1, ROLE>(a, cx, i, it, st, t, r4, u4, v4);
            }
        } else if constexpr (NST == S - 1) {
            // position 0 after step K enters window K (its d slots and the ring row of this parity; the other wave fills the other one)
            if constexpr (K <= RW) window_push_dir<DOWN>(st.w[K - 1], qin);
            else deep_window_push<DOWN>(W, lane, it, st.d0[L], qin);
        }
    }
}

// One iteration: position i takes step 1, position i - 1 step 2, ..., position i - (D - 1) step D (stored).  NST = number of stages
// that have a row: 1..D-1 in iterations 0..D-2 (code of their own, i a constant: the pipeline fills, the two waves of the pair hand
// over), D in the loop.  PFD = 1: `cur` holds position i on entry and position i + 1 is gathered into `nxt` first; the caller swaps
// the two from one iteration to the next.  PAR >= 0: the parity of i as a constant (the LDS ring slots become immediate offsets).
// ROLE (k_deep2): a FRONT wave is a march of depth D whose last stage hands its row to the back wave instead of storing it; a BACK wave
// is a march of depth D whose "step 1" is that row, already in `cur` (D, NST, i: the wave's own).  Neither has barriers of its own.
template <int BC, bool MASK, bool MACRO, int D, int RW, int PFD, bool DOWN, int NST, int PAR = -1, int ROLE = DEEP_WHOLE>
__device__ __forceinline__ void deep_iter(const StepArgs &a, const DeepCtx &cx, const int i_, DeepState<RW, D - 1 - RW> &st, Row1 &cur,
                                          Row1 &nxt)
{
    // (the mailbox -- the last LDS window -- is read at the top of iteration K <= RW and takes its own first row later in iteration D - 2)
    static_assert(RW >= 0 && RW <= 2 && D - 1 - RW >= 1 && (ROLE == DEEP_WHOLE ? D - 2 > RW : D - 2 >= RW),
                  "register windows hand over through the last LDS window while it is idle");
    static_assert((D - 1 - RW) * DEEP_WSLOTS <= 40, "a wave has 40 KB of LDS at four waves per CU");
    const int lane = cx.lane, x4 = cx.x4;
    const int i = NST < D ? NST - 1 : i_;
    const int it = PAR >= 0 ? PAR : i;              // (only its parity is used)
    auto row_at = [&](int p) { return DOWN ? cx.ym - 1 - p : cx.ym + p; };
    // ---- what the other wave published for "position -1" of the register windows in its previous iteration ---------------
    f4a (*MB)[64] = cx.mine + (D - 2 - RW) * DEEP_WSLOTS;       // my last LDS window, idle until iteration D - 2: the mailbox
    if (RW >= 1 && NST == 2) { st.w[0].g2 = MB[5][lane]; st.w[0].g5 = MB[6][lane]; st.w[0].g6 = MB[7][lane]; }
    if (RW >= 2 && NST == 3) { st.w[RW >= 2 ? 1 : 0].g2 = MB[2][lane]; st.w[RW >= 2 ? 1 : 0].g5 = MB[3][lane]; st.w[RW >= 2 ? 1 : 0].g6 = MB[4][lane]; }
    // ---- step 1 of position i (from memory) --------------------------------------------------------------------------------
    // (behind the last position the last row is gathered again -- a cache hit that nobody consumes: no condition on i)
#ifdef LB_DIAG
    if (!((a.diag & (1 << 23)) && i > 0))
#endif
    if constexpr (ROLE != DEEP_BACK) {
        if (ROLE == DEEP_FRONT) {
            deep_row_take_lds<BC, MASK>(cx.dma, lane, cur);
            deep_row_issue_lds<BC, MASK>(a, row_at(min(i + 1, cx.n_iter - 1)), x4, cx.dma_off, nxt);
        } else if (PFD && LB_DEEP_MANUAL) {
            deep_row_take<BC, MASK, (NST < D ? 0 : (MACRO ? 12 : 9))>(cur);
            deep_row_issue<BC, MASK>(a, row_at(min(i + 1, cx.n_iter - 1)), x4, nxt);
        } else if (PFD) deep_row_load<BC, MASK>(a, row_at(min(i + 1, cx.n_iter - 1)), x4, nxt);
        else deep_row_load<BC, MASK>(a, row_at(i), x4, cur);
    }
    f4a (&q1)[9] = cur.q;
    f4a r4, u4, v4;
    const uc4 mk = cur.mk;
    if (ROLE != DEEP_BACK && cur.have) {
        gather_merge<BC, true>(a, x4, q1, cur.wp);
        LB_DEEP_NOCOLLIDE collide_row<BC, MASK, true>(a, x4, a.y0 + cur.rr, q1, mk, r4, u4, v4);
    }
    if (NST == 1) {
        if constexpr (RW >= 1) deep_publish<DOWN>(cx.other + (D - 2 - RW) * DEEP_WSLOTS, lane, 5, q1);      // (mailbox)
        else deep_publish<DOWN>(cx.other, lane, 2 + 3 * 1, q1);
    }
    deep_stage<BC, MASK, MACRO, D, RW, DOWN, NST, 2, ROLE>(a, cx, i, it, st, q1, r4, u4, v4);
    if (MASK) st.mhist = ((st.mhist | mask_word(mk)) << 1) & (0x01010101u * (unsigned)(((1 << D) - 2) & 0xff));
    if (ROLE == DEEP_WHOLE && NST < D) __syncthreads();     // what was published in this iteration is consumed in the next
}

// the filling iterations 0..D-2, one after the other (NST = 1..D-1); PFD = 1: the two row buffers swap roles every iteration
template <int BC, bool MASK, bool MACRO, int D, int RW, int PFD, bool DOWN, int NST>
__device__ __forceinline__ void deep_fill(const StepArgs &a, const DeepCtx &cx, DeepState<RW, D - 1 - RW> &st, Row1 &ra, Row1 &rb)
{
    if constexpr (NST < D) {
        if (PFD == 1 && (NST & 1) == 0) deep_iter<BC, MASK, MACRO, D, RW, PFD, DOWN, NST>(a, cx, NST - 1, st, rb, ra);
        else deep_iter<BC, MASK, MACRO, D, RW, PFD, DOWN, NST>(a, cx, NST - 1, st, ra, PFD ? rb : ra);
        deep_fill<BC, MASK, MACRO, D, RW, PFD, DOWN, NST + 1>(a, cx, st, ra, rb);
    }
}

// One wave's march: columns [x0, x0 + 256) of which all but the skirt lanes at either end are stored, `len` rows from the pair's middle line `ym`
// upward or downward; len + D - 1 iterations.
template <int BC, bool MASK, bool MACRO, int D, int RW, int PFD, bool DOWN>
__device__ __forceinline__ void deep_march(const StepArgs &a, const int x0, const int ym, const int len, f4a (*mine)[64],
                                           f4a (*other)[64])
{
    DeepCtx cx;
    cx.lane = threadIdx.x;
    const int xr = x0 + cx.lane * 4;                 // true column of my first cell: -8 .. ; may lie beyond either end of the box
    // lanes beyond an end of the box: the periodic images as far as the skirt reaches (behind it: the last image lane's lines), or
    // -- walls -- copies of the lane at that end (a wall column's rule rebuilds whatever it pulled from outside)
    constexpr int SKL = deep_skirt_lanes(D);
    if (BC == LB_BC_PERIODIC) cx.x4 = xr < 0 ? xr + a.nx : (xr >= a.nx ? (xr - a.nx < 4 * SKL ? xr - a.nx : 4 * (SKL - 1)) : xr);
    else cx.x4 = min(max(xr, 0), (a.nx - 1) & ~3);
    cx.store_lane = cx.lane >= SKL && cx.lane <= 63 - SKL && xr < a.nx;
    cx.ym = ym; cx.n_iter = len + D - 1;
    cx.mine = mine; cx.other = other;
    DeepState<RW, D - 1 - RW> st = {};
    auto row_at = [&](int p) { return DOWN ? ym - 1 - p : ym + p; };
    Row1 ra, rb;
    if (PFD && LB_DEEP_MANUAL) deep_row_issue<BC, MASK>(a, row_at(0), cx.x4, ra);
    else if (PFD) deep_row_load<BC, MASK>(a, row_at(0), cx.x4, ra);
    deep_fill<BC, MASK, MACRO, D, RW, PFD, DOWN, 1>(a, cx, st, ra, rb);
    // (the steady iterations wait with the count of a steady iteration's stores: the first one has none behind it)
    if (PFD && LB_DEEP_MANUAL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (PFD == 1 && !deep_pairs(MASK, BC)) {
        // one iteration per trip; the row gathered ahead moves into place (position D - 1 is in ra or rb by its parity)
        if ((D - 1) & 1) ra = rb;
        for (int i = D - 1; i < cx.n_iter; ++i) {
            deep_iter<BC, MASK, MACRO, D, RW, PFD, DOWN, D>(a, cx, i, st, ra, rb);
            ra = rb;
        }
    } else if (PFD == 1) {
        // position i is in ra for even i, in rb for odd i; the steady iterations in pairs
        constexpr int P0 = (D - 1) & 1;
        int i = D - 1;
        for (; i + 1 < cx.n_iter; i += 2) {
            deep_iter<BC, MASK, MACRO, D, RW, PFD, DOWN, D, P0>(a, cx, i, st, P0 ? rb : ra, P0 ? ra : rb);
            deep_iter<BC, MASK, MACRO, D, RW, PFD, DOWN, D, 1 - P0>(a, cx, i + 1, st, P0 ? ra : rb, P0 ? rb : ra);
        }
        if (i < cx.n_iter) deep_iter<BC, MASK, MACRO, D, RW, PFD, DOWN, D, P0>(a, cx, i, st, P0 ? rb : ra, P0 ? ra : rb);
    } else {
        for (int i = D - 1; i < cx.n_iter; ++i) deep_iter<BC, MASK, MACRO, D, RW, PFD, DOWN, D>(a, cx, i, st, ra, ra);
    }
    if (PFD && LB_DEEP_MANUAL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the row gathered behind the last position)
}

// Launch geometry as k_step5 / k_step6: one workgroup = one segment pair of one strip (two waves), XCD-transposed order, shorter
// segments for the two wall-column strips.  LDS: (D - 1 - RW) x 8 KiB per wave.
// (LB_DEEP_OCC2: timing probes only -- two workgroups per SIMD pair where the depth's registers and LDS allow it, e.g. D = 4:
//  tools/r06/occ2_probe.sh, profiles/r06_experiments.txt section 5)
#ifndef LB_DEEP_OCC2
#define LB_DEEP_OCC2 0
#endif
#ifndef LB_DEEP_KATTR
#define LB_DEEP_KATTR
#endif
template <int BC, bool MASK, bool MACRO, int D, int RW, int PFD>
__global__ __launch_bounds__(64 * STEP4_WAVES, ((PFD && !LB_DEEP_OCC2) ? 1 : 2)) LB_DEEP_KATTR void k_deep(const StepArgs a, int strips, int seg_rows, int nsegs, int row_end)
{
    __shared__ f4a lds_win[STEP4_WAVES][(D - 1 - RW) * DEEP_WSLOTS][64];
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int item = xcd_item(blockIdx.x, gridDim.x);
#ifdef LB_DIAG
    const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz
#endif
    int sx, sy;
    if (item < strips * nsegs) {
        sx = item % strips;
        sy = item / strips;
    } else {
        if (!a.edge_seg_rows) return;
        const int j = item - strips * nsegs;
        sx = (j & 1) ? strips - 1 : 0;
        sy = nsegs + (j >> 1);
    }
    int stride = a.seg_stride;
    if (a.edge_seg_rows && (sx == 0 || sx == strips - 1)) stride = seg_rows = a.edge_seg_rows;
    const int ya = a.row_begin + sy * stride;
    if (ya >= row_end) return;                          // (both waves of the workgroup: the barriers stay matched)
    const int yb = min(ya + seg_rows, row_end);
    const int ym = ya + (yb - ya) / 2;                  // the pair's middle line: wave 0 marches down from it, wave 1 up
    const int x0 = sx * deep_valid(D) - 4 * deep_skirt_lanes(D);
    // (Round 6 let a workgroup whose rows hold no solid cell in its strip -- two scalar loads of per-strip running counts kept by the
    //  host -- take the march WITHOUT the obstacle swap, inside the same kernel: its loop is 235 instructions per row shorter (2381
    //  against 2616 in a walled kernel), and it bought nothing: the reference's 3751 x 1251 pipe + disc 256 against 259 k MLUPS, pipe +
    //  disc 4096^2 331 / 329, config 5's image 331 / 329, cavity + disc 6144^2 382 against 397 k (-4 %), periodic + random mask (no clean
    //  workgroup) 423 / 423 -- the kernel grows from 26 to 46 thousand instructions, and two CUs whose four workgroups are not all of one
    //  kind run 80 KB of steady loops through a 64 KB instruction cache.  profiles/r06_experiments.txt section 4; commit 64f91cf has the code.)
    if (wy == 0) deep_march<BC, MASK, MACRO, D, RW, PFD, true>(a, x0, ym, ym - ya, lds_win[0], lds_win[1]);
    else deep_march<BC, MASK, MACRO, D, RW, PFD, false>(a, x0, ym, yb - ym, lds_win[1], lds_win[0]);
#ifdef LB_DIAG
    if ((a.diag & 4096) && threadIdx.x == 0) {
        // per-wave timeline into the (otherwise unused) rho array: start, end (100 MHz ticks), XCC id, HW id, item, rows (tools/wave_timeline.py)
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        unsigned *o = reinterpret_cast<unsigned *>(a.rho) + 8 * (item * STEP4_WAVES + wy);
        o[0] = (unsigned)diag_t0; o[1] = (unsigned)(diag_t0 >> 32); o[2] = (unsigned)t1; o[3] = (unsigned)(t1 >> 32);
        o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
        o[6] = (unsigned)(item * STEP4_WAVES + wy); o[7] = (unsigned)(wy ? yb - ym : ym - ya);
    }
#endif
}

}  // namespace
