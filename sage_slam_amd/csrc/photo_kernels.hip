// photo_kernels.hip -- fused feature-metric (photometric) linearize / error kernels for gfx950.
//
// Replaces cuda/photometric_factor_kernels.cpp:33-368 (+ host reduction :1061-1164) and :370-522
// (+ :990-1059) of the reference.  The reference writes every residual's Jacobian row to global memory
// ([L,N,FS,13+CS], 236 MB per dense 128x160 edge) and reduces it with two GEMMs; here nothing per-residual
// ever leaves the CU:
//
//   per source pixel n (one lane):   G = sum_l w_l sum_c h h^T (2x2),  v = sum_l w_l sum_c h r,  e
//        with h = (fx_l gx, fy_l gy) the level-scaled sampled gradient, r = m (f0 - f1)
//   J_c = h^T P_unit,  P_unit = Q M_n,  Q = [A | q] (2x7): A = Jpi_unit R1^T [I | -[Xw]x],  q = dpi/dd
//        M_n = [[I6, -I6, 0, 0], [0, 0, s0 b_n^T, d_n/s0]]            (P_pose1 = -P_pose0)
//   => per pixel S = Q^T G Q (7x7), u = Q^T v (7); 37 scalars are summed with wave64 DPP reductions,
//      the code blocks  sum_n S66 b_n b_n^T  and  sum_n [S(0:6,6); S66 d; u6] b_n^T  are f32 MFMA
//      (v_mfma_f32_16x16x4_f32, K = 4 pixels) contractions fed from the LDS basis tile.
//
// Algorithmic bytes per source pixel: 4*[4*FS*rho + CS + 6].
#include <type_traits>

#include "sage_device.h"
#include "sage_internal.h"
#include "finalize_bodies.h"

namespace sage
{

struct PhotoParams
{
  PhotoEdge single;
  const PhotoEdge *table;
  const WorkItem *work;
  float *partials;
  SagePyramid pyr;
  float w[SAGE_MAX_LEVELS];
  float eps;
  int tiles_per_block; // consecutive kTile-pixel sub-tiles one workgroup accumulates before writing its partial
  int width, height;   // level-0 size as integers (scalar values for the basis descriptor)
  float rx[SAGE_MAX_LEVELS], ry[SAGE_MAX_LEVELS]; // fx_l/fx_0, fy_l/fy_0 (host-computed: no per-level divisions on the device)
  int lw[SAGE_MAX_LEVELS], lh[SAGE_MAX_LEVELS];   // level sizes as integers
  float geo_loss_param; // error kernel: > 0 -> also the geometric error of the edge (LaunchCommon::fused_geo_loss_param)
  int n_work;
  // linearize: a workgroup writes one partial record per `flush` sub-tiles (record index = rec_first[edge] + tile / flush);
  // flush == tiles_per_block is the plain "one record per work item"
  const int32_t *rec_first;
  int flush;
  float merge_w; // > 0: merged linearize (LaunchCommon::merge_geo_weight) -- the geometric factor weight w_g
  // r06: a pyramid whose per-level focal ratios are not exact powers of two (pyramid_is_dyadic) -- the level coordinate is then
  // formed per pixel exactly as the reference writes it, ((p + 0.5) * fx_l) / fx_0 - 0.5 (photometric_factor_kernels.cpp:101-103,
  // :142-144), on the texture-path sampler; dyadic pyramids (every CameraPyramid of even level sizes) keep the host quotient
  int exact_coord;
};

__device__ __forceinline__ int load_loc(const void *loc, int is64, int n)
{
  return is64 ? (int)reinterpret_cast<const long long *>(loc)[n] : reinterpret_cast<const int *>(loc)[n];
}

#ifndef SAGE_PHOTO_ERR_WAVES
#define SAGE_PHOTO_ERR_WAVES 5 // error pass at FS = 16: 5 workgroups per CU (94 VGPRs, no spills; 4: 0.255 ms, 5: 0.246, 6 spills: 0.45);
                               // FS = 32 lands at 102 VGPRs = 5 per SIMD on its own (asking for it costs 2 spills)
#endif
#ifndef SAGE_PHOTO_WAVES
#define SAGE_PHOTO_WAVES 3 // workgroups per CU the linearize kernel is register-budgeted for (x4 waves)
#endif
// error pass: channel groups of a level unrolled together (their tap loads are then in flight together).  Fully unrolled,
// the FS = 32 kernel (8 groups) needs 168 VGPR + 61 spilled registers; 4 -> 150 / 0
#ifndef SAGE_PHOTO_ERR_GUNROLL
#define SAGE_PHOTO_ERR_GUNROLL 4
#endif
#ifndef SAGE_PHOTO_PRIO_SAMPLING
#define SAGE_PHOTO_PRIO_SAMPLING 3 // s_setprio of the linearize kernel's sampling phase (its contraction phases run at 0)
#endif
// r06: the next sub-tile's per-pixel inputs of phase A (location, homogeneous coordinates, depth: a chain of dependent global
// loads, 3.3 k cycles of a wave's 36 k per sub-tile in the wave timeline) are asked for during the current sub-tile's phases
// C / D: photometric linearize 0.622 -> 0.611 ms in the cold micro-bench (profiles/r06_kernel_ab_experiments.txt); +4 VGPRs.
// 0 = off (A/B)
#ifndef SAGE_PHOTO_PREFETCH_A
#define SAGE_PHOTO_PREFETCH_A 1
#endif
#ifndef SAGE_PHOTO_LOCKSTEP
// r06: the four waves of a workgroup issue every staging fill TOGETHER (one s_barrier ahead of each fill; a wave on the
// texture path or with a dead slice executes the same number of barriers).  The waves' tiles are x-neighbours: a box row
// of 11 texels spans 2.4 cache lines of which 1.3 also belong to the neighbour's box -- issued within a few hundred
// cycles of each other the second request hits the CU's L1 instead of going to the L2 (which holds ~5 us of this
// kernel's stream and never caught it).  Config 4 (FS = 32): L1 -> L2 requests 9.5e7 -> 7.7e7, memory-side fetch
// -8 %, linearize 1.20 -> 1.105 ms, error pass 0.625 -> 0.528 ms; K = 64 (FS = 16): -1 % on all three kernels
// (profiles/r06_kernel_ab_experiments.txt s12).  Bit mask: 1 linearize FS >= 32, 2 linearize FS = 16, 4 / 8 error pass.
#define SAGE_PHOTO_LOCKSTEP 15
#endif
#ifndef SAGE_PHOTO_FS32_SLOAD
// FS >= 32 (BASELINE config 4): r05 kept the per-lane pose loads of phase C and one group of basis prefetch there -- bound by its memory
// side (fetch 1.58 x algorithmic), that kernel lost 4 % with the scalar loads + six groups in flight that FS = 16 runs with.  With the
// lockstep fills (fetch 1.38 x) the order is reversed: scalar loads + six groups -1.7 % (profiles/r06_kernel_ab_experiments.txt s16).
#define SAGE_PHOTO_FS32_SLOAD 1 // FS >= 32: poses of phase C by per-lane loads (0, r05) or scalar loads (1)
#endif
#ifndef SAGE_PHOTO_FS32_AHEAD
#define SAGE_PHOTO_FS32_AHEAD 6 // FS >= 32: pixel groups of basis rows in flight ahead of the contraction (r05: 1)
#endif
#ifndef SAGE_PHOTO_LIN_GUNROLL
#define SAGE_PHOTO_LIN_GUNROLL 8
#endif
// scripts/isa_census.py compiles this file with -DSAGE_PHASE_MARKERS: comment lines in the generated code that name the
// phase which follows (a scheduling barrier on both sides keeps the instructions of a phase between its markers)
#ifdef SAGE_PHASE_MARKERS
#define SAGE_PHASE(name)                                 \
  do                                                     \
  {                                                      \
    __builtin_amdgcn_sched_barrier(0);                   \
    asm volatile("; SAGE_PHASE " name ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)
#else
#define SAGE_PHASE(name) \
  do                     \
  {                      \
  } while (0)
#endif

// scripts/photo_trace.py builds a variant with -DSAGE_PHOTO_TRACE: lane 0 of every wave of the merged linearize stamps
// s_memtime at the phase boundaries of every sub-tile into a static device array (+ a header: HW_ID, XCC_ID, s_memrealtime),
// read back through sage_debug_photo_trace().  Diagnostic only -- never defined in the product build.
#ifdef SAGE_PHOTO_TRACE
constexpr int kTraceMaxWg = 3072, kTraceSubs = 9, kTraceMarks = 8; // [wg][wave][sub (8 = header)][mark]
__device__ unsigned long long g_photo_trace[(size_t)kTraceMaxWg * 4 * kTraceSubs * kTraceMarks];
#define SAGE_TMARK(m)                                                                                          \
  do                                                                                                           \
  {                                                                                                            \
    if constexpr (JAC && MODE == 2)                                                                            \
    {                                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      const unsigned long long t_ = __builtin_readcyclecounter();                                              \
      if (bid < kTraceMaxWg && sub < kTraceSubs - 1 && (threadIdx.x & 63) == 0)                                \
        g_photo_trace[(((size_t)bid * 4 + (threadIdx.x >> 6)) * kTraceSubs + sub) * kTraceMarks + (m)] = t_;   \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
    }                                                                                                          \
  } while (0)
#else
#define SAGE_TMARK(m) \
  do                  \
  {                   \
  } while (0)
#endif

// per-pixel hand-over from the sampling phase (lane = pixel) to the contraction phase (lane = (channel i, pixel k)):
//   [0..7] rows of the cross tile: c(6) = S(0:6,6), sigma*d, u6      [8] sigma = S66     [9] loc*CS*4 (int bits)
//   [10..23] A rows of the pose tile: G*Q rows (2 x 6), v (2)         [24..37] its B columns: Q (2 x 6), q6*d (2)
// Operand rows / columns 14, 15 of the pose tile read whatever follows (finite values): they only reach the output
// rows / columns 14, 15, which nothing reads.
constexpr int kPhotoStashLD = 40; // floats per pixel, 16-byte aligned rows

// ---- LDS-staged sampler of the linearize kernel (engine layout) ------------------------------------------------------
// A wave's 64 source pixels (an 8 x 8 image tile when the samples are in the engine's tile order) warp to a compact
// footprint in the destination keyframe: per channel group the wave copies the texels of that footprint's bounding box
// -- level 0 into one region, the coarser levels together into a second one, for feat1, d/dx and d/dy -- into its own
// stash memory (idle during the sampling phase) with LDS-direct buffer loads (lane = texel, no VGPR round trip), and
// every bilinear tap is a ds_read_b128 instead of a gather through the CU's texture path: ~9 wave-loads per channel
// group through the texture path instead of 48.  The regions of group g + 1 are filled while group g is reduced
// (level 0 right after its taps were read, the coarse levels at the end of the group).  Footprints that do not fit
// (strong zoom / rotation between the keyframes, samples that are not in tile order) take the texture path.
constexpr int kStageLevels = 4;  // the staged sampler is built for 4-level pyramids (straight-line code, static counters)
// (capacities = whole rounds of 64 lanes: the number of loads per fill is fixed -- 6 and 3 -- and the wait counts exact)
constexpr int kStageCap0 = 128;  // texels per array, level 0: two rounds      (11 x 11 footprints fit)
constexpr int kStageCapC = 64;   // texels per array, levels >= 1 together: one round (6 x 6 + 4 x 4 + 3 x 3 = 61)
static_assert(3 * (kStageCap0 + kStageCapC) * 16 <= 64 * kPhotoStashLD * 4, "the staging regions alias the wave's stash");
// error pass: only feat1 is sampled -- one array, kErrStageGroups channel groups staged together (their regions side by
// side: [group][level 0: kStageCap0 | coarse: kStageCapC] float4 = 3 KiB per group and wave)
#ifndef SAGE_PHOTO_ERR_STAGE_GROUPS
#define SAGE_PHOTO_ERR_STAGE_GROUPS 2
#endif
constexpr int kErrStageGroups = SAGE_PHOTO_ERR_STAGE_GROUPS;
constexpr int kErrStageGroupBytes = (kStageCap0 + kStageCapC) * 16;

// 16-byte LDS read at a byte address of the workgroup's LDS allocation (ds_read_b128 v, vaddr offset:imm)
__device__ __forceinline__ f32x4 lds_read16(uint32_t addr)
{
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const f32x4 __attribute__((address_space(3))) *LdsF4;
  return *(LdsF4)(addr);
#else
  (void)addr;
  return f32x4{0.f, 0.f, 0.f, 0.f};
#endif
}

// one LDS-direct 16-byte-per-lane load: LDS[lds_byte + 16 * lane] = buffer[voff + soff]  (lds_byte, soff wave-uniform).
// Invisible to the compiler's wait-count bookkeeping: every consumer sits behind an explicit vm_wait below.
// s_nop 4: the hazard recogniser does not look inside inline asm -- an SGPR operand written by a VALU instruction right
// before the statement (v_readlane of a spilled SGPR, v_readfirstlane) needs 5 wait states before a VMEM instruction
// reads it (the first build faulted on exactly that), and m0 needs one before the LDS-direct load.
template <uint32_t LDS_OFF>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, uint32_t lds_base, uint32_t voff, uint32_t soff)
{
  // (m0 = base + compile-time offset formed by the instruction itself: nine live SGPRs of precomputed addresses less)
  asm volatile("s_add_u32 m0, %0, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_base), "v"(voff), "s"(r), "s"(soff), "n"(LDS_OFF)
               : "memory", "scc");
}
// 16-byte global load (wave-uniform base, per-lane byte offset) the compiler does not track either: the value must pass
// through vm_wait_keep before its first use (scripts/check_asm_loads.py verifies that nothing touches the registers earlier)
__device__ __forceinline__ f32x4 gload16(const float *sbase, uint32_t voff)
{
  f32x4 v;
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
  return v;
}
// basis rows of the contraction phase: same discipline (the compiler batches tracked loads four at a time and waits for the
// first right behind the fourth -- two pixel groups of cover; as hand-tracked loads they stay SAGE_PHOTO_AHEAD groups ahead)
__device__ __forceinline__ f32x2 bload8(__amdgpu_buffer_rsrc_t r, uint32_t voff)
{
  f32x2 v;
  asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(r));
  return v;
}
__device__ __forceinline__ float bload4(__amdgpu_buffer_rsrc_t r, uint32_t voff)
{
  float v;
  asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(r));
  return v;
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
  if constexpr (I < N)
  {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
template <int N>
__device__ __forceinline__ void vm_wait_keep2(f32x2 &v)
{
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N));
}
template <int N>
__device__ __forceinline__ void vm_wait_keep(f32x4 &v)
{
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N) : "memory");
}
__device__ __forceinline__ void lgkm_wait0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// optimisation barrier on six running sums: what feeds them is computed before this point, what reads them after it
__device__ __forceinline__ void pin6(f32x2 &a, f32x2 &b, f32x2 &c, f32x2 &d, f32x2 &e, f32x2 &f)
{
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}

// Two world-from-keyframe poses (wave-uniform addresses) through the scalar cache: eight s_load issued together, ONE wait.
// (r05: written as loads of a uniform address + v_readfirstlane the compiler made them per-lane flat loads through one
// register quad -- seven serialised L2 round trips and 21 VALU slots per 64-pixel slice.)  The pointers are not assumed
// contiguous (the per-edge operators pass separate tensors).
__device__ __forceinline__ void sload_pose_pair(const float *R0, const float *t0, const float *R1, const float *t1, Pose &p0,
                                                Pose &p1)
{
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  f32x8 a8, b8;
  f32x2 at2, bt2;
  float a1, b1, at1, bt1;
  asm volatile("s_load_dwordx8 %0, %8, 0x0\n\t"
               "s_load_dword %1, %8, 0x20\n\t"
               "s_load_dwordx2 %2, %9, 0x0\n\t"
               "s_load_dword %3, %9, 0x8\n\t"
               "s_load_dwordx8 %4, %10, 0x0\n\t"
               "s_load_dword %5, %10, 0x20\n\t"
               "s_load_dwordx2 %6, %11, 0x0\n\t"
               "s_load_dword %7, %11, 0x8\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(a8), "=&s"(a1), "=&s"(at2), "=&s"(at1), "=&s"(b8), "=&s"(b1), "=&s"(bt2), "=&s"(bt1)
               : "s"(R0), "s"(t0), "s"(R1), "s"(t1)
               : "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i)
  {
    p0.R[i] = a8[i];
    p1.R[i] = b8[i];
  }
  p0.R[8] = a1;
  p1.R[8] = b1;
  p0.t[0] = at2[0]; p0.t[1] = at2[1]; p0.t[2] = at1;
  p1.t[0] = bt2[0]; p1.t[1] = bt2[1]; p1.t[2] = bt1;
#else
  p0 = load_pose2(R0, t0);
  p1 = load_pose2(R1, t1);
#endif
}

// level-l pixel coordinate of a level-0 coordinate (:101-103, :142-144): ONE rounding sequence for the lanes and for the
// bounding box (floor() of it is then monotone in p, so the box of [min p, max p] contains every lane's taps)
__device__ __forceinline__ float level_coord(float p, float ratio) { return __builtin_fmaf(p + 0.5f, ratio, -0.5f); }
// the reference's own expression (true multiplication, true division): non-dyadic pyramids
__device__ __forceinline__ float level_coord_exact(float p05, float fl, float f0) { return (p05 * fl) / f0 - 0.5f; }

// wave64 float min / max, returned wave-uniform.  The DPP ladder of wave_sum with the min / max fused into the DPP
// instruction itself: v = op(dpp(v), v), a lane whose DPP source is masked or out of range keeps its value.  Written as
// asm because fminf / fmaxf through __builtin_amdgcn_update_dpp costs a v_mov_dpp, two canonicalising v_max and the
// operation per step (100 VALU instructions for the four reductions of a sub-tile; 24 like this).  s_nop 1: a DPP read
// of a VGPR the previous VALU instruction wrote needs two wait states, and the assembler does not insert them.
#define SAGE_DPP_STEP(OP, CTRL) asm("s_nop 1\n\t" OP " %0, %0, %0 " CTRL : "+v"(v))
template <bool IS_MIN>
__device__ __forceinline__ float wave_fminmax(float v)
{
  if (IS_MIN)
  {
    SAGE_DPP_STEP("v_min_f32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    SAGE_DPP_STEP("v_min_f32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    SAGE_DPP_STEP("v_min_f32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf");
    SAGE_DPP_STEP("v_min_f32_dpp", "row_mirror row_mask:0xf bank_mask:0xf");
    SAGE_DPP_STEP("v_min_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf");
    SAGE_DPP_STEP("v_min_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf");
  }
  else
  {
    SAGE_DPP_STEP("v_max_f32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    SAGE_DPP_STEP("v_max_f32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    SAGE_DPP_STEP("v_max_f32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf");
    SAGE_DPP_STEP("v_max_f32_dpp", "row_mirror row_mask:0xf bank_mask:0xf");
    SAGE_DPP_STEP("v_max_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf");
    SAGE_DPP_STEP("v_max_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf");
  }
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#undef SAGE_DPP_STEP

// second-level accumulators of the noise-critical tiles (the two cross tiles and the pose tile: 3 x 4 floats per lane),
// one region per wave (see "second level" in the kernel)
#ifndef SAGE_PHOTO_L2_TILES
#define SAGE_PHOTO_L2_TILES 3
#endif
constexpr int kPhotoL2Tiles = SAGE_PHOTO_L2_TILES;
// CS = 32: the noise-critical tiles (two cross tiles, pose tile) accumulate the even and the odd pixel groups of a sub-tile
// in two accumulator sets -- fp32 chains of 32 instead of 64 fmaf, a quarter of the chain's rounding variance -- merged
// before the second-level update (r03: K = 64 LM step vs the fp32 oracle 9.1 -> 7.6e-5 together with a record every 4
// sub-tiles; same register allocation, kernel time unchanged)
#ifndef SAGE_PHOTO_ALT_ACC
#define SAGE_PHOTO_ALT_ACC 1
#endif

// One (level, channel-group) step of the sampler in the engine's channel-group layout: 4 taps x (f1, gx, gy) dwordx4
// loads + the pre-sampled source features.
template <bool JAC>
struct TapBatch
{
  f32x4 t1[4], tx[JAC ? 4 : 1], ty[JAC ? 4 : 1], f0;
};

// MODE 0: reference layout [FS][P] / [2][FS][P], dword gathers, source features sampled in-kernel (per-edge operator API)
// MODE 2: MODE 1 + the merged linearize (the geometric edge's code0 blocks ride in this kernel's contractions; linearize only)
// MODE 1: engine layout [FS/4][P][4] (one dwordx4 per tap and channel group, 1 KiB contiguous per wave) with the
//         source features pre-sampled per keyframe; the (level, group) steps are software-pipelined: the 13 loads of
//         step it+1 are in flight while step it is reduced (two register batches).
//
// Phases per 64-pixel wave slice (no workgroup barrier inside the sub-tile loop, the waves only meet for the final sum):
//   A  warp: depth from the keyframe's depth map, projection, mask                      (lane = pixel)
//   B  sampling: G, v, e accumulated over levels and channels                           (lane = pixel)
//   C  per-pixel 7x7 reduced system; rows for the contractions -> wave-private LDS stash.  The 34 pose / scale
//      scalar sums (Q^T G Q, Q^T G q6 d, Q^T v, q6^T v d) are one more MFMA tile (A = [GQ0 GQ1 v], B = [Q0 Q1 q6 d]):
//      a DPP reduction + LDS accumulate per scalar costs ~15 VALU and a serialised LDS round trip each
//      (36 x per 64 pixels was a fifth of the kernel's instruction stream); err, n_inliers and sigma d^2 are
//      accumulated per lane and reduced once per workgroup
//   D  code blocks: f32 MFMA 16x16x4 with the basis rows loaded from global memory directly in operand layout
//      (lane = (channel pair i, pixel k): 16 lanes x dwordx2 = one 128-byte basis row; CS = 32: operand block 0 = even
//      channels, block 1 = odd channels)
template <int CS, int FS, bool JAC, int MODE>
__global__ __launch_bounds__(kBlock, !JAC ? (FS == 16 ? SAGE_PHOTO_ERR_WAVES : 3) : SAGE_PHOTO_WAVES) void photo_kernel(const PhotoParams prm)
{
  constexpr bool PACKED = MODE >= 1;
  constexpr bool MERGE = JAC && MODE == 2; // engine layout + the geometric edge's code0 blocks (LaunchCommon::merge_geo_weight)
  constexpr int NB = CS / 16;
  constexpr int NG = FS / 4;
  // channel groups of a level unrolled together (linearize: all; error pass: at most SAGE_PHOTO_ERR_GUNROLL)
  constexpr int GUNROLL_MAX = JAC ? SAGE_PHOTO_LIN_GUNROLL : SAGE_PHOTO_ERR_GUNROLL;
  constexpr int GUNROLL = NG > GUNROLL_MAX ? GUNROLL_MAX : NG;
  constexpr int NT = photo_tiles(CS);
  constexpr int YY = NT; // the pose tile: accumulated like the code tiles, folded into the scalar slots at the end
  constexpr int STASH = JAC ? kWaves * 64 * kPhotoStashLD : 1;
  constexpr int SUMBUF = JAC ? (NT + 1) * 256 : 1;
  __shared__ __attribute__((aligned(16))) float s_mem[STASH > SUMBUF ? STASH : SUMBUF];
  __shared__ float s_red[kWaves * 4]; // per wave: linearize {sigma d^2, error, inliers}, error pass {error, inliers, geo error, inliers}
  // second level of the noise-critical tiles: [wave][tile][r][lane]
  __shared__ float s_l2[JAC ? kWaves * kPhotoL2Tiles * 256 : 1];
  __shared__ __attribute__((aligned(16))) float s_estage[(!JAC && PACKED) ? kWaves * kErrStageGroups * kErrStageGroupBytes / 4 : 4];

  const int tid_wg = threadIdx.x, tid = tid_wg, lane = tid & 63, wave = tid_wg >> 6;
#ifdef SAGE_PHOTO_TRACE
  const unsigned long long t_entry_ = __builtin_readcyclecounter(); // first instruction of the wave: dispatch vs prologue
#endif
  const int bid = (int)blockIdx.x;
  WorkItem wi = prm.work[bid];
  wi.edge = uni(wi.edge);
  wi.tile = uni(wi.tile);
  PhotoEdge E = prm.table ? prm.table[wi.edge] : prm.single;
  E.feat0 = uni(E.feat0); E.feat1 = uni(E.feat1); E.grad1 = uni(E.grad1); E.dpt0 = uni(E.dpt0);
  E.feat1_pk = uni(E.feat1_pk);
  E.basis0 = uni(E.basis0); E.mask1 = uni(E.mask1); E.homo = uni(E.homo); E.loc = uni(E.loc);
  E.R0 = uni(E.R0); E.t0 = uni(E.t0); E.R1 = uni(E.R1); E.t1 = uni(E.t1); E.R10 = uni(E.R10); E.t10 = uni(E.t10);
  E.N = uni(E.N); E.loc_is_i64 = uni(E.loc_is_i64); E.f0s = uni(E.f0s);
  E.dpt1_geo = uni(E.dpt1_geo);
  E.geo_px = uni(E.geo_px);
  const int N = E.N;

  // ---- poses (wave-uniform) ----
  Pose p10;
  if (E.R10)
    p10 = load_pose2(E.R10, E.t10);
  else
  {
    p10 = relative_pose(load_pose2(E.R0, E.t0), load_pose2(E.R1, E.t1));
    // computed on the vector unit (no scalar float arithmetic on this part), wave-uniform: moved to SGPRs -- as 12 long-lived
    // VGPRs it was spilled to scratch and reloaded (with a full vmcnt drain) in every sub-tile
#pragma unroll
    for (int i = 0; i < 9; ++i)
      p10.R[i] = uni(p10.R[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
      p10.t[i] = uni(p10.t[i]);
  }
  if (JAC && PACKED)
  {
    // r05: the linearize has register room (140 of the 168 VGPRs three waves per SIMD allow) and no SGPRs left -- the 12 values
    // lived in spilled SGPRs and cost 24 v_readlane + their wait states per slice (warp in phase A, again in phase C): as
    // (opaque) VGPRs they are plain operands
#pragma unroll
    for (int i = 0; i < 9; ++i)
      asm volatile("" : "+v"(p10.R[i]));
#pragma unroll
    for (int i = 0; i < 3; ++i)
      asm volatile("" : "+v"(p10.t[i]));
  }

  // per-lane constants of the staged sampler's box computation (see "the 16 coordinates" below): lane 4 l + c, c = x0, x1, y0, y1
  float box_ratio = 0.f;
  int box_hi = 0;
  if (PACKED)
  {
    const int bc = lane & 3;
#pragma unroll
    for (int l = 0; l < kStageLevels; ++l)
      if ((lane >> 2) == l || (l == kStageLevels - 1 && (lane >> 2) >= kStageLevels))
      {
        box_ratio = bc < 2 ? prm.rx[l] : prm.ry[l];
        box_hi = (bc < 2 ? prm.lw[l] : prm.lh[l]) - 1 + (bc & 1);
      }
  }
  const SagePyramid &pyr = prm.pyr;
  const float fx0 = pyr.cam[0].fx, fy0 = pyr.cam[0].fy, cx0 = pyr.cam[0].cx, cy0 = pyr.cam[0].cy;
  const int W0 = prm.width, H0 = prm.height;
  const uint32_t pyr_bytes = (uint32_t)FS * (uint32_t)pyr.P * 4u;
  const __amdgpu_buffer_rsrc_t r_f0 = make_rsrc(E.feat0, pyr_bytes);
  // engine layout: feat1 | fx_l d/dx | fy_l d/dy of the destination keyframe are consecutive [3][FS/4][P][4] arrays under
  // one descriptor (array a at + a * pyr_bytes, through the scalar offset)
  const __amdgpu_buffer_rsrc_t r_f1 = make_rsrc(PACKED ? E.feat1_pk : E.feat1, (PACKED && JAC) ? 3u * pyr_bytes : pyr_bytes);
  const __amdgpu_buffer_rsrc_t r_g1 = make_rsrc((JAC && !PACKED) ? E.grad1 : E.feat1, (JAC && !PACKED) ? 2u * pyr_bytes : pyr_bytes);
  const __amdgpu_buffer_rsrc_t r_b0 = make_rsrc(E.basis0, (uint32_t)W0 * (uint32_t)H0 * (uint32_t)(CS * 4));
  const uint32_t plane = (uint32_t)pyr.P * 4u;
  const int nlev = pyr.levels;

  if (tid < kWaves * 4)
    s_red[tid] = 0.f;
  f32x4 acc[NT + 1];
#pragma unroll
  for (int t = 0; t < NT + 1; ++t)
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float err_acc = 0.f, vm_acc = 0.f, sdd_acc = 0.f; // lane-local sums over the sub-tiles: error, inliers, sigma d^2
  float gerr_acc = 0.f;                             // error kernel, fused geometric error
  const bool fuse_geo = !JAC && prm.geo_loss_param > 0.f && E.dpt1_geo != nullptr;
  const float geo_loss = E.geo_loss > 0.f ? E.geo_loss : prm.geo_loss_param; // per-link parameter (mapper.cpp:369)
  if (!JAC && PACKED && tid == 0)
    s_estage[0] = 0.f; // (the staging memory is otherwise only reached through LDS-direct loads and integer addresses: an
                       //  array nothing in the program writes or reads is not allocated)
  __syncthreads(); // s_red zeroed


#ifdef SAGE_PHOTO_TRACE
  if constexpr (JAC && MODE == 2)
  {
    if (bid < kTraceMaxWg && lane == 0)
    {
      unsigned long long *h = g_photo_trace + (((size_t)bid * 4 + wave) * kTraceSubs + (kTraceSubs - 1)) * kTraceMarks;
      h[0] = (unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11));  // HW_REG_HW_ID
      h[1] = (unsigned)__builtin_amdgcn_s_getreg(20 | (31 << 11)); // HW_REG_XCC_ID
      h[2] = __builtin_amdgcn_s_memrealtime();
      h[3] = __builtin_readcyclecounter();
      h[4] = (unsigned)wi.edge;
      h[5] = (unsigned)wi.tile;
      h[6] = t_entry_;
    }
  }
#endif

  const int nsub = min(prm.tiles_per_block, (N + kTile - 1) / kTile - wi.tile);
  const int flush = JAC ? max(1, prm.flush) : 1;
  const int rec_base = (JAC && prm.rec_first) ? uni(prm.rec_first[wi.edge]) + wi.tile / flush : bid;
  int run_pos = 0, rec_idx = 0; // sub-tiles since the last partial record, records written so far
#if SAGE_PHOTO_PREFETCH_A
  // next sub-tile's per-pixel inputs of phase A, asked for during the contraction phase of the current one (linearize, engine layout)
  int pf_loc = 0;
  float pf_d = 1.0f, pf_hm0 = 0.f, pf_hm1 = 0.f, pf_hm2 = 1.f;
  bool pf_in = false;
#endif
  for (int sub = 0; sub < nsub; ++sub)
  {
  // (the thread index is re-derived per sub-tile behind an opaque barrier: everything computed from it -- operand lane
  //  offsets, stash addresses, record indices -- would otherwise be hoisted out of this loop as invariants and spilled)
  int tid = tid_wg;
  if (JAC)
    asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = tid >> 6;
  float *const st_w = s_mem + wave * 64 * kPhotoStashLD; // this wave's stash
  if (JAC)
  {
    // the second-level tiles are zero at this point (moved to their LDS slot after every sub-tile, or flushed): said
    // explicitly, they are dead between the sub-tiles' contraction phases instead of 12 registers held (spilled) across
    // the sampling phase
#pragma unroll
    for (int t = NT + 1 - kPhotoL2Tiles; t < NT + 1; ++t)
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  SAGE_PHASE("A_warp");
  SAGE_TMARK(0);
  const int tile = wi.tile + sub;
  const int n = tile * kTile + tid;
  bool in_range = n < N;
  int my_loc;
  float d;
  float hm[3] = {0.f, 0.f, 1.f};
#if SAGE_PHOTO_PREFETCH_A
  if (JAC && PACKED && sub > 0)
  {
    in_range = pf_in;
    my_loc = pf_loc;
    d = pf_d;
    hm[0] = pf_hm0; hm[1] = pf_hm1; hm[2] = pf_hm2;
  }
  else
#endif
  {
  my_loc = in_range ? load_loc(E.loc, E.loc_is_i64, n) : 0;
  // a location outside the image is dropped here (the window engine and sage_sort_locations reject it up front; the
  // reference's tensor index() would throw): no out-of-bounds read of the depth map / basis rows
  in_range = in_range && (unsigned)my_loc < (unsigned)(W0 * H0);
  my_loc = in_range ? my_loc : 0;
  // depth of the source pixel: s0*(bias + basis.code), read from the keyframe's depth map (:1094-1095)
  d = in_range ? E.dpt0[my_loc] : 1.0f;
  if (in_range)
  {
    hm[0] = E.homo[3 * n + 0];
    hm[1] = E.homo[3 * n + 1];
    hm[2] = E.homo[3 * n + 2];
  }
  }
  float rh[3], X[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
  {
    rh[i] = p10.R[i * 3 + 0] * hm[0] + p10.R[i * 3 + 1] * hm[1] + p10.R[i * 3 + 2] * hm[2];
    X[i] = d * rh[i] + p10.t[i];
  }
  const bool pos = X[2] > prm.eps; // photometric_factor_kernels.cpp:96
  // true divisions, like the reference: the coordinate chain's fp32 rounding is the dominant noise term of
  // the whole linearisation (every residual of the pixel inherits it), so no reciprocal shortcut here.
  const float p = (X[0] / X[2]) * fx0 + cx0; // :142-144 (level-0 pixel coordinates)
  const float q = (X[1] / X[2]) * fy0 + cy0;
  const float m = mask_lookup(E.mask1, p, q, W0, H0);
  float vm = (pos && in_range) ? m : 0.0f; // sampled_valid_mask_1 (:237)
  if (fuse_geo)
  {
    // geometric_factor_kernels.cpp:127-218 at the same warp: D1 bilinear at the level-0 coordinates (no half-pixel
    // shift), rho = D1 - z, Cauchy error log(1 + (m rho)^2 / c) for the pixels in front of the camera
    Taps tg;
    make_taps(tg, p, q, W0, H0);
    float Ds = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      Ds += tg.w[k] * E.dpt1_geo[tg.off[k]];
    const float mr = m * (Ds - X[2]);
    gerr_acc += (pos && in_range) ? logf(1.0f + mr * mr / geo_loss) : 0.f;
  }

  float G00 = 0.f, G01 = 0.f, G11 = 0.f, v0 = 0.f, v1 = 0.f, err = 0.f;
  // one (level, channel group) step: bilinear interpolation of the 4 taps of feat1 (and d/dx, d/dy), difference to the
  // pre-sampled source quad, the six sums of the level (:200-236) as channel PAIRS -- every update is one v_pk_fma_f32
  // on naturally aligned register pairs of the interpolated quads
  auto reduce_step = [&](const TapBatch<JAC> &B, const float (&tw)[4], f32x2 &q00, f32x2 &q01, f32x2 &q11, f32x2 &qa0,
                         f32x2 &qa1, f32x2 &qee) {
    f32x4 f1 = {0.f, 0.f, 0.f, 0.f}, gx = f1, gy = f1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      f1 += tw[k] * B.t1[k];
      if (JAC)
      {
        gx += tw[k] * B.tx[k];
        gy += tw[k] * B.ty[k];
      }
    }
    const f32x4 d4 = B.f0 - f1;
    const f32x2 dl = {d4[0], d4[1]}, dh = {d4[2], d4[3]};
    qee += dl * dl;
    qee += dh * dh;
    if (JAC)
    {
      const f32x2 xl = {gx[0], gx[1]}, xh = {gx[2], gx[3]}, yl = {gy[0], gy[1]}, yh = {gy[2], gy[3]};
      q00 += xl * xl;
      q00 += xh * xh;
      q01 += xl * yl;
      q01 += xh * yh;
      q11 += yl * yl;
      q11 += yh * yh;
      qa0 += xl * dl;
      qa0 += xh * dh;
      qa1 += yl * dl;
      qa1 += yh * dh;
    }
  };
  // the level's weight and its fx_l / fy_l scaling (h = (fx_l gx, fy_l gy)) applied once to the six sums of a step
  auto fold_step = [&](int l, const f32x2 &q00, const f32x2 &q01, const f32x2 &q11, const f32x2 &qa0, const f32x2 &qa1,
                       const f32x2 &qee) {
    // engine layout: the packed pyramids (and the pre-sampled source features) carry sqrt(w_l) and the gradient pyramids
    // the level's focal lengths, h = sqrt(w_l) (fx_l d/dx, fy_l d/dy): every product is weighted already
    const float wl = PACKED ? 1.0f : prm.w[l];
    err += wl * (qee[0] + qee[1]);
    if (JAC)
    {
      if (PACKED)
      {
        G00 += q00[0] + q00[1];
        G01 += q01[0] + q01[1];
        G11 += q11[0] + q11[1];
        v0 += qa0[0] + qa0[1];
        v1 += qa1[0] + qa1[1];
      }
      else
      {
        const float fxl = pyr.cam[l].fx, fyl = pyr.cam[l].fy;
        const float wx = wl * fxl, wy = wl * fyl;
        G00 += (wx * fxl) * (q00[0] + q00[1]);
        G01 += (wx * fyl) * (q01[0] + q01[1]);
        G11 += (wy * fyl) * (q11[0] + q11[1]);
        v0 += wx * (qa0[0] + qa0[1]);
        v1 += wy * (qa1[0] + qa1[1]);
      }
    }
  };
  bool slice_live = true; // linearize, engine layout: false when no pixel of this wave's slice is an inlier
  SAGE_PHASE("B_setup");
  SAGE_TMARK(1);
  if (PACKED)
  {
    // wave priority: a wave in its sampling phase goes ahead of the waves of its SIMD that are in their VALU / MFMA
    // phases (r03: 0.906 -> 0.879 ms)
    if (JAC)
      __builtin_amdgcn_s_setprio(SAGE_PHOTO_PRIO_SAMPLING);
    bool staged = false;
    float pmn = 0.f, pmx = 0.f, qmn = 0.f, qmx = 0.f;
    {
      // bounding box of the inliers' level-0 destination coordinates (an inlier passed the mask lookup: it lies inside
      // the image).  A slice without inliers contributes exact zeros to every sum: it is skipped.
      const bool lv = vm != 0.f;
      const float inf = __builtin_inff();
      pmn = wave_fminmax<true>(lv ? p : inf);
      pmx = wave_fminmax<false>(lv ? p : -inf);
      qmn = wave_fminmax<true>(lv ? q : inf);
      qmx = wave_fminmax<false>(lv ? q : -inf);
      slice_live = pmn <= pmx;
    }
    // per level: box origin, width, first slot inside its staging region (wave-uniform)
    int bx0[kStageLevels] = {}, by0[kStageLevels] = {}, bwd[kStageLevels] = {}, bhd[kStageLevels] = {}, sb[kStageLevels] = {};
    int cnt0 = 0, cntC = 0;
    if (slice_live && nlev == kStageLevels && !prm.exact_coord)
    {
      // (floor coordinate of the first tap .. second tap, NOT clamped to the image: columns -1 / W_l and rows -1 / H_l
      //  are part of the box when an inlier's taps reach them -- those taps carry weight 0 and are filled with a
      //  repeated border texel -- so that every inlier's four taps sit at a0, a0 + 16, a0 + row, a0 + row + 16):
      //    x0 = min(max(floor(c_l(pmn)), -1), W_l - 1)      x1 = max(min(floor(c_l(pmx)) + 1, W_l), x0 + 1)     (y alike)
      // r05: the 16 coordinates (4 levels x {x0, x1, y0, y1}) are formed in ONE pass, lane 4 l + c taking coordinate c of
      // level l with its ratio and upper clamp from two per-lane constants (box_ratio, box_hi) -- as uniform values on the
      // vector unit they cost ~100 of a slice's VALU slots.  Same operations per value, same integers.  (The lower clamp -1
      // is harmless for x1 / y1: a value below it is replaced by x0 + 1 >= 0 either way.)
      const int bc = lane & 3, odd = bc & 1;
      const float bsel = (bc & 2) ? (odd ? qmx : qmn) : (odd ? pmx : pmn);
      int bw = min(max((int)floorf(level_coord(bsel, box_ratio)) + odd, -1), box_hi);
      bw = max(bw, __builtin_amdgcn_update_dpp(0, bw, 0xA0, 0xf, 0xf, false) + odd); // (x1, y1 read x0, y0: quad_perm [0,0,2,2])
#pragma unroll
      for (int l = 0; l < kStageLevels; ++l)
      {
        const int x0 = __builtin_amdgcn_readlane(bw, 4 * l + 0), x1 = __builtin_amdgcn_readlane(bw, 4 * l + 1);
        const int y0 = __builtin_amdgcn_readlane(bw, 4 * l + 2), y1 = __builtin_amdgcn_readlane(bw, 4 * l + 3);
        bx0[l] = x0;
        by0[l] = y0;
        bwd[l] = x1 - x0 + 1;
        bhd[l] = y1 - y0 + 1;
        const int c = bwd[l] * bhd[l];
        if (l == 0)
        {
          sb[l] = 0;
          cnt0 = c;
        }
        else
        {
          sb[l] = cntC;
          cntC += c;
        }
      }
      staged = cnt0 <= kStageCap0 && cntC <= kStageCapC; // (6 x 6 + 4 x 4 + 3 x 3 = 61 coarse texels, one more row / column at an image border)
    }
    constexpr bool LOCKSTEP = ((SAGE_PHOTO_LOCKSTEP) >> ((JAC ? 0 : 2) + (FS >= 32 ? 0 : 1))) & 1;
    constexpr int NBAR = JAC ? NG : (NG + kErrStageGroups - 1) / kErrStageGroups; // staging fills per slice
    if (!slice_live)
    {
      // nothing to sample
      if constexpr (LOCKSTEP)
        for (int g = 0; g < NBAR; ++g)
          __builtin_amdgcn_s_barrier();
    }
    else if (staged)
    {
      // ================= LDS-staged sampler =================
      // linearize: the wave's stash holds [3 arrays][kStageCap0] float4 of level 0, then [3][kStageCapC] of the coarse levels;
      // error pass: its own region, per staged group [kStageCap0 | kStageCapC] float4 of feat1
      const uint32_t lds0 = JAC ? uni((int)(uint32_t)(uintptr_t)st_w)
                                : uni((int)(uint32_t)(uintptr_t)(s_estage + wave * (kErrStageGroups * kErrStageGroupBytes / 4)));
      const uint32_t ldsC = lds0 + (JAC ? 3u : 1u) * kStageCap0 * 16u;
      // lane -> texel of the two regions (two rounds of 64 slots each), as byte offsets inside a channel group's plane
      uint32_t vo0[2], voC;
      {
        const float ib0 = __builtin_amdgcn_rcpf((float)bwd[0]);
#pragma unroll
        for (int r = 0; r < 2; ++r)
        {
          const int t = min(lane + 64 * r, cnt0 - 1);
          const int by = (int)(((float)t + 0.5f) * ib0);
          const int bx = t - by * bwd[0];
          const int yy = min(max(by0[0] + by, 0), prm.lh[0] - 1), xx = min(max(bx0[0] + bx, 0), prm.lw[0] - 1);
          vo0[r] = (uint32_t)(pyr.level_offsets[0] + yy * prm.lw[0] + xx) * 16u;
        }
        {
          const int t = min(lane, cntC - 1);
          int s_ = sb[1], w_ = bwd[1], x_ = bx0[1], y_ = by0[1], lo_ = pyr.level_offsets[1], lw_ = prm.lw[1], lh_ = prm.lh[1];
#pragma unroll
          for (int l = 2; l < kStageLevels; ++l)
          {
            const bool up = t >= sb[l];
            s_ = up ? sb[l] : s_;
            w_ = up ? bwd[l] : w_;
            x_ = up ? bx0[l] : x_;
            y_ = up ? by0[l] : y_;
            lo_ = up ? pyr.level_offsets[l] : lo_;
            lw_ = up ? prm.lw[l] : lw_;
            lh_ = up ? prm.lh[l] : lh_;
          }
          const int tl = t - s_;
          const int by = (int)(((float)tl + 0.5f) * __builtin_amdgcn_rcpf((float)w_));
          const int bx = tl - by * w_;
          const int yy = min(max(y_ + by, 0), lh_ - 1), xx = min(max(x_ + bx, 0), lw_ - 1);
          voC = (uint32_t)(lo_ + yy * lw_ + xx) * 16u;
        }
      }
      // pre-sampled source quads [L][NG][N][4]: base of (level, group) wave-uniform, lane offset n * 16
      const uint32_t f0_vo = (uint32_t)(in_range ? n : 0) * 16u;
      // (the 16 (level, group) bases are formed at their loads with scalar arithmetic from a laundered N: as loop invariants the
      //  compiler keeps all of them in SGPRs across the sub-tile loop, spills them into VGPR lanes and pays two
      //  v_readlane per load -- 32 VALU slots per sub-tile)
      auto f0_base = [&](int l, int g) {
        unsigned nq = (unsigned)N;
        asm volatile("" : "+s"(nq)); // (volatile: stays next to the load that uses the base)
        return E.f0s + (size_t)((unsigned)(l * NG + g) * nq) * 4;
      };
      // per level and lane: the 4 tap weights and the LDS byte address a0 of the first tap (xf, yf); the others are
      // (xc, yf) = a0 + 16, (xf, yc) = a0 + row, (xc, yc) = a0 + row + 16 with row = 16 * box width (wave-uniform).  Inliers
      // are box-interior by construction; the clamp only matters for the other lanes (wild coordinates, weight x 0).  A
      // tap outside the image has weight 0 and reads a finite repeated border texel.
      float tw[kStageLevels][4];
      uint32_t tap[kStageLevels];
#pragma unroll
      for (int l = 0; l < kStageLevels; ++l)
      {
        Taps td;
        make_taps(td, level_coord(p, prm.rx[l]), level_coord(q, prm.ry[l]), prm.lw[l], prm.lh[l]);
        const int cx0 = min(max(td.xf - bx0[l], 0), bwd[l] - 2), cy0 = min(max(td.yf - by0[l], 0), bhd[l] - 2);
        tap[l] = (l == 0 ? lds0 : ldsC) + (uint32_t)(sb[l] + cy0 * bwd[l] + cx0) * 16u;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          tw[l][k] = td.w[k];
      }
      if constexpr (JAC)
      {
      // fill a region with channel group g: always 6 loads (3 arrays x 2 rounds; slots past the box repeat its last
      // texel) for level 0 and 3 for the coarse levels -- the wait counts below are exact
      auto stage0 = [&](uint32_t soff) {
        dma16<0u>(r_f1, lds0, vo0[0], soff);
        dma16<kStageCap0 * 16u>(r_f1, lds0, vo0[0], soff + pyr_bytes);
        dma16<2u * kStageCap0 * 16u>(r_f1, lds0, vo0[0], soff + 2u * pyr_bytes);
        dma16<1024u>(r_f1, lds0, vo0[1], soff);
        dma16<kStageCap0 * 16u + 1024u>(r_f1, lds0, vo0[1], soff + pyr_bytes);
        dma16<2u * kStageCap0 * 16u + 1024u>(r_f1, lds0, vo0[1], soff + 2u * pyr_bytes);
      };
      auto stageC = [&](uint32_t soff) {
        dma16<3u * kStageCap0 * 16u>(r_f1, lds0, voC, soff);
        dma16<3u * kStageCap0 * 16u + kStageCapC * 16u>(r_f1, lds0, voC, soff + pyr_bytes);
        dma16<3u * kStageCap0 * 16u + 2u * kStageCapC * 16u>(r_f1, lds0, voC, soff + 2u * pyr_bytes);
      };
      SAGE_PHASE("B_taps");
  SAGE_TMARK(2);
      lgkm_wait0(); // the stash reads of the previous sub-tile's contraction are done before the region is rewritten
      // the source quads of a channel group live in a ring of four registers quads: quad l is reloaded with the next
      // group's level l right after its use, a full group (~4 level steps) before it is needed
      f32x4 f0q[kStageLevels];
#pragma unroll
      for (int l = 0; l < kStageLevels; ++l)
        f0q[l] = gload16(f0_base(l, 0), f0_vo);
      if constexpr (LOCKSTEP)
        __builtin_amdgcn_s_barrier();
      stage0(0u);
      stageC(0u);
      // running sums of the slice as channel PAIRS (one v_pk_fma_f32 per sum and step; the halves meet once, below)
      f32x2 P00 = {0.f, 0.f}, P01 = P00, P11 = P00, Pv0 = P00, Pv1 = P00, Pee = P00;
      auto level_step = [&](auto lc, const f32x4 &f0v) {
        constexpr int l = decltype(lc)::value;
        constexpr uint32_t AS = (l == 0 ? kStageCap0 : kStageCapC) * 16u; // byte stride between the three arrays
        const uint32_t a0 = tap[l], a2 = a0 + (uint32_t)bwd[l] * 16u;
        // two batches (feat1 + d/dx, then d/dy): 8 + 4 reads in flight instead of 12
        f32x4 t1[4], tx[4], ty[4];
        t1[0] = lds_read16(a0);           tx[0] = lds_read16(a0 + AS);
        t1[1] = lds_read16(a2 + 16u);     tx[1] = lds_read16(a2 + 16u + AS);
        t1[2] = lds_read16(a2);           tx[2] = lds_read16(a2 + AS);
        t1[3] = lds_read16(a0 + 16u);     tx[3] = lds_read16(a0 + 16u + AS);
        // nd = sum_k w_k t_k - f0 (the residual's subtraction rides in the interpolation chain: r = -nd; the chain starts
        // from the pre-sampled quad, which the producer stores negated -- r05: 2 v_xor per step less)
        f32x4 nd = f0v, gx = {0.f, 0.f, 0.f, 0.f}, gy = gx; // (f0s holds the NEGATED source features)
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
          nd += tw[l][k] * t1[k];
          gx += tw[l][k] * tx[k];
        }
        __builtin_amdgcn_sched_barrier(0);
        ty[0] = lds_read16(a0 + 2u * AS);
        ty[1] = lds_read16(a2 + 16u + 2u * AS);
        ty[2] = lds_read16(a2 + 2u * AS);
        ty[3] = lds_read16(a0 + 16u + 2u * AS);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          gy += tw[l][k] * ty[k];
        const f32x2 dl = {nd[0], nd[1]}, dh = {nd[2], nd[3]};
        const f32x2 xl = {gx[0], gx[1]}, xh = {gx[2], gx[3]}, yl = {gy[0], gy[1]}, yh = {gy[2], gy[3]};
        // the texels carry sqrt(w_l): the six sums take their weighted products directly (one v_pk_fma_f32 each)
        P00 += xl * xl;
        P00 += xh * xh;
        P01 += xl * yl;
        P01 += xh * yh;
        P11 += yl * yl;
        P11 += yh * yh;
        Pv0 -= xl * dl;
        Pv0 -= xh * dh;
        Pv1 -= yl * dl;
        Pv1 -= yh * dh;
        Pee += dl * dl;
        Pee += dh * dh;
      };
      // Straight-line over the channel groups (static wait counts, and no control-flow merge while a register is still in
      // flight: the compiler may place a register copy at a merge, ahead of the wait -- scripts/check_asm_loads.py walks
      // the generated code for exactly that).  Issue order of the loads the waits count:
      //   f0(0,0..3) L0(0) LC(0) | per group g: [level 0] f0(g+1,0) L0(g+1) [level 1] f0(g+1,1) [level 2] f0(g+1,2)
      //   [level 3] f0(g+1,3) LC(g+1)
      // level 0 of group g needs f0(g,0) and L0(g): younger are f0(g,1..3) and the 3 loads of LC(g) (g = 0: only LC(0));
      // level 1 needs LC(g) (f0(g,1) is older): younger are f0(g+1,0) and the 6 loads of L0(g+1) (nothing in the last
      // group); levels 2, 3 are covered by the wait of level 1 (vmcnt(15) = no wait, ties the register to it).  pin6 keeps every step where it is written: without a use
      // at that point the reduction sinks towards the end and the taps of several levels stay live.
#pragma unroll
      for (int g = 0; g < NG; ++g)
      {
        const bool more = g + 1 < NG; // (compile-time after unrolling)
        const uint32_t soffn = (uint32_t)(g + 1) * plane * 4u;
        if (g == 0)
          vm_wait_keep<3>(f0q[0]);
        else
          vm_wait_keep<6>(f0q[0]);
        level_step(std::integral_constant<int, 0>{}, f0q[0]);
        pin6(P00, P01, P11, Pv0, Pv1, Pee);
        if (more)
        {
          f0q[0] = gload16(f0_base(0, g + 1), f0_vo);
          lgkm_wait0(); // level-0 taps have been read: the region takes the next group
          if constexpr (LOCKSTEP)
            __builtin_amdgcn_s_barrier();
          stage0(soffn);
          vm_wait_keep<7>(f0q[1]);
        }
        else
          vm_wait_keep<0>(f0q[1]);
        level_step(std::integral_constant<int, 1>{}, f0q[1]);
        pin6(P00, P01, P11, Pv0, Pv1, Pee);
        if (more)
        {
          f0q[1] = gload16(f0_base(1, g + 1), f0_vo);
          vm_wait_keep<15>(f0q[2]);
        }
        else
          vm_wait_keep<0>(f0q[2]);
        level_step(std::integral_constant<int, 2>{}, f0q[2]);
        pin6(P00, P01, P11, Pv0, Pv1, Pee);
        if (more)
        {
          f0q[2] = gload16(f0_base(2, g + 1), f0_vo);
          vm_wait_keep<15>(f0q[3]);
        }
        else
          vm_wait_keep<0>(f0q[3]);
        level_step(std::integral_constant<int, 3>{}, f0q[3]);
        pin6(P00, P01, P11, Pv0, Pv1, Pee);
        if (more)
        {
          f0q[3] = gload16(f0_base(3, g + 1), f0_vo);
          lgkm_wait0();
          stageC(soffn);
        }
      }
      G00 = P00[0] + P00[1];
      G01 = P01[0] + P01[1];
      G11 = P11[0] + P11[1];
      v0 = Pv0[0] + Pv0[1];
      v1 = Pv1[0] + Pv1[1];
      err = Pee[0] + Pee[1];
      SAGE_PHASE("B_end");
  SAGE_TMARK(3);
      }
      else
      {
      // ---- error pass: feat1 only.  kErrStageGroups channel groups are staged together (3 loads each: two rounds of level 0,
      //      one of the coarse levels) and reduced level by level; no hand-over between batches beyond the single wait (this
      //      kernel runs 4+ waves per SIMD, the other waves cover it)
      f32x2 Pee = {0.f, 0.f};
#pragma unroll
      for (int g0 = 0; g0 < NG; g0 += kErrStageGroups)
      {
        lgkm_wait0(); // (the previous batch's taps / the previous sub-tile's have been read)
        if constexpr (LOCKSTEP)
          __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < kErrStageGroups; ++j)
        {
          const uint32_t soff = (uint32_t)(g0 + j) * plane * 4u;
          const uint32_t base = lds0 + (uint32_t)j * kErrStageGroupBytes;
          dma16<0u>(r_f1, base, vo0[0], soff);
          dma16<1024u>(r_f1, base, vo0[1], soff);
          dma16<kStageCap0 * 16u>(r_f1, base, voC, soff);
        }
        f32x4 f0v[kStageLevels][kErrStageGroups];
#pragma unroll
        for (int l = 0; l < kStageLevels; ++l)
#pragma unroll
          for (int j = 0; j < kErrStageGroups; ++j)
            f0v[l][j] = gload16(f0_base(l, g0 + j), f0_vo);
        // everything of this batch has landed (quads and regions)
#pragma unroll
        for (int l = 0; l < kStageLevels; ++l)
#pragma unroll
          for (int j = 0; j < kErrStageGroups; ++j)
            vm_wait_keep<0>(f0v[l][j]);
#pragma unroll
        for (int l = 0; l < kStageLevels; ++l)
        {
          const uint32_t a0 = tap[l], a2 = a0 + (uint32_t)bwd[l] * 16u;
          f32x2 qee = {0.f, 0.f};
#pragma unroll
          for (int j = 0; j < kErrStageGroups; ++j)
          {
            const uint32_t o = (uint32_t)j * kErrStageGroupBytes;
            const f32x4 t0 = lds_read16(a0 + o), t1 = lds_read16(a2 + 16u + o), t2 = lds_read16(a2 + o), t3 = lds_read16(a0 + 16u + o);
            f32x4 nd = f0v[l][j]; // (negated at the producer)
            nd += tw[l][0] * t0;
            nd += tw[l][1] * t1;
            nd += tw[l][2] * t2;
            nd += tw[l][3] * t3;
            const f32x2 dl = {nd[0], nd[1]}, dh = {nd[2], nd[3]};
            qee += dl * dl;
            qee += dh * dh;
          }
          Pee += qee; // (the texels carry sqrt(w_l))
        }
      }
      err = Pee[0] + Pee[1];
      }
    }
    else
    {
      // ================= texture-path sampler: (level, channel group) steps, the 13 dwordx4 loads of a step issued
      // together, then reduced =================
      SAGE_PHASE("B_texture_path");
      if constexpr (LOCKSTEP)
        for (int g = 0; g < NBAR; ++g)
          __builtin_amdgcn_s_barrier();
      const f32x4 *f0s = reinterpret_cast<const f32x4 *>(E.f0s) + (in_range ? n : 0);
      for (int l = 0; l < nlev; ++l)
      {
        Taps td;
        if (prm.exact_coord) // (wave-uniform)
          make_taps(td, level_coord_exact(p + 0.5f, pyr.cam[l].fx, fx0), level_coord_exact(q + 0.5f, pyr.cam[l].fy, fy0), prm.lw[l],
                    prm.lh[l]);
        else
          make_taps(td, level_coord(p, prm.rx[l]), level_coord(q, prm.ry[l]), prm.lw[l], prm.lh[l]);
        const uint32_t lo = (uint32_t)pyr.level_offsets[l];
        uint32_t dof[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          dof[k] = (lo + (uint32_t)td.off[k]) * 16u;
        f32x2 q00 = {0.f, 0.f}, q01 = q00, q11 = q00, qa0 = q00, qa1 = q00, qee = q00;
#pragma unroll GUNROLL
        for (int g = 0; g < NG; ++g)
        {
          const uint32_t soff = (uint32_t)g * plane * 4u;
          TapBatch<JAC> B;
          B.f0 = -f0s[((size_t)l * NG + g) * N]; // (stored negated for the staged sampler's FMA chain)
#pragma unroll
          for (int k = 0; k < 4; ++k)
          {
            B.t1[k] = buf_load4(r_f1, dof[k], soff);
            if (JAC)
            {
              B.tx[k] = buf_load4(r_f1, dof[k], soff + pyr_bytes);
              B.ty[k] = buf_load4(r_f1, dof[k], soff + 2u * pyr_bytes);
            }
          }
          __builtin_amdgcn_sched_barrier(0); // without it hipcc serialises load->wait->use through one register quad
          reduce_step(B, td.w, q00, q01, q11, qa0, qa1, qee);
        }
        fold_step(l, q00, q01, q11, qa0, qa1, qee);
      }
    }
  }
  else
  {
    // source coordinates at level 0 (+0.5): from homo in the Jacobian kernel (:101-103), from loc1d in the
    // error-only kernel (:423-424, :1012-1014)
    float su, sv;
    if (JAC)
    {
      su = hm[0] * fx0 + cx0 + 0.5f;
      sv = hm[1] * fy0 + cy0 + 0.5f;
    }
    else
    {
      su = (float)(my_loc % W0) + 0.5f;
      sv = (float)(my_loc / W0) + 0.5f;
    }
    for (int l = 0; l < nlev; ++l)
    {
      const float fxl = pyr.cam[l].fx, fyl = pyr.cam[l].fy;
      const int Wl = prm.lw[l], Hl = prm.lh[l];
      const float rx = prm.rx[l], ry = prm.ry[l];
      Taps ts, td;
      if (prm.exact_coord) // (wave-uniform)
      {
        make_taps(ts, level_coord_exact(su, fxl, fx0), level_coord_exact(sv, fyl, fy0), Wl, Hl);
        make_taps(td, level_coord_exact(p + 0.5f, fxl, fx0), level_coord_exact(q + 0.5f, fyl, fy0), Wl, Hl);
      }
      else
      {
        make_taps(ts, su * rx - 0.5f, sv * ry - 0.5f, Wl, Hl);
        make_taps(td, (p + 0.5f) * rx - 0.5f, (q + 0.5f) * ry - 0.5f, Wl, Hl);
      }
      const uint32_t lo = (uint32_t)pyr.level_offsets[l];
      uint32_t so[4], dof[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
      {
        so[k] = (lo + (uint32_t)ts.off[k]) * 4u;
        dof[k] = (lo + (uint32_t)td.off[k]) * 4u;
      }
      float g00 = 0.f, g01 = 0.f, g11 = 0.f, a0 = 0.f, a1 = 0.f, ee = 0.f;
#pragma unroll 4
      for (int c = 0; c < FS; ++c)
      {
        const uint32_t soff = (uint32_t)c * plane;
        float f0 = 0.f, f1 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
          f0 += ts.w[k] * buf_load(r_f0, so[k], soff);
          f1 += td.w[k] * buf_load(r_f1, dof[k], soff);
        }
        const float diff = f0 - f1;
        ee += diff * diff;
        if (JAC)
        {
          const uint32_t soff_y = (uint32_t)(FS + c) * plane;
          float gx = 0.f, gy = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k)
          {
            gx += td.w[k] * buf_load(r_g1, dof[k], soff);
            gy += td.w[k] * buf_load(r_g1, dof[k], soff_y);
          }
          const float hx = fxl * gx, hy = fyl * gy;
          g00 += hx * hx;
          g01 += hx * hy;
          g11 += hy * hy;
          a0 += hx * diff;
          a1 += hy * diff;
        }
      }
      const float wl = prm.w[l];
      err += wl * ee;
      if (JAC)
      {
        G00 += wl * g00;
        G01 += wl * g01;
        G11 += wl * g11;
        v0 += wl * a0;
        v1 += wl * a1;
      }
    }
  }
  err *= vm; // within_mask * pow(diff,2)  (:228)
  err_acc += err;
  vm_acc += vm;
  if (!JAC)
    continue;

  // ---- per-pixel 7x7 reduced system ----
  SAGE_PHASE("C_rows");
  __builtin_amdgcn_s_setprio(0);
#if SAGE_PHOTO_PREFETCH_A
  if (PACKED && sub + 1 < nsub)
  {
    const int n2 = n + kTile;
    pf_in = n2 < N;
    pf_loc = pf_in ? load_loc(E.loc, E.loc_is_i64, n2) : 0;
    const float *hp2 = E.homo + 3 * (pf_in ? n2 : 0);
    pf_hm0 = hp2[0]; pf_hm1 = hp2[1]; pf_hm2 = hp2[2];
  }
#endif
  if (slice_live)
  {
  const bool live = vm != 0.0f;
  // merged linearize (LaunchCommon::merge_geo_weight): the geometric kernel's hand-over for this pixel, asked for first
  f32x4 gp = {0.f, 0.f, 0.f, 0.f};
  if (MERGE)
    gp = reinterpret_cast<const f32x4 *>(E.geo_px)[in_range ? n : 0];
  const float vm2 = vm * vm; // gradient and residual both carry m (:200, :234)
  G00 *= vm2;
  G01 *= vm2;
  G11 *= vm2;
  v0 *= vm2;
  v1 *= vm2;
  float Q[2][7];
  float dXz[6]; // z-row of dX/dT0 (merged linearize: the geometric edge's pose row starts from it)
  {
    // world-from-keyframe poses, re-read per sub-tile through the scalar cache: held across the sampling phase their 24 SGPRs
    // were spilled to VGPR lanes and every use paid a v_readlane
    Pose p0, p1;
    if constexpr (FS >= 32 && !SAGE_PHOTO_FS32_SLOAD)
    {
      // (r05, FS = 32 while its fetch was 1.58 x the algorithmic bytes: 4 % faster with the per-lane loads the compiler makes of this --
      //  seven round trips in series that hold the wave back from its next burst of requests; off since the lockstep fills)
      const float *R0p = E.R0, *R1p = E.R1;
      asm volatile("" : "+s"(R0p), "+s"(R1p));
      p0 = load_pose2(R0p, E.t0);
      p1 = load_pose2(R1p, E.t1);
    }
    else
      sload_pose_pair(E.R0, E.t0, E.R1, E.t1, p0, p1);
    // (engine layout: the homogeneous coordinates are read again and the warp of phase A recomputed -- same operations,
    //  same values -- instead of ten registers staying live across the sampling phase)
    if (PACKED)
    {
      const float *hp = E.homo + 3 * (in_range ? n : 0);
      hm[0] = in_range ? hp[0] : 0.f;
      hm[1] = in_range ? hp[1] : 0.f;
      hm[2] = in_range ? hp[2] : 1.f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
      {
        rh[i] = p10.R[i * 3 + 0] * hm[0] + p10.R[i * 3 + 1] * hm[1] + p10.R[i * 3 + 2] * hm[2];
        X[i] = d * rh[i] + p10.t[i];
      }
    }
    const float inv_z = 1.0f / X[2];
    float Xw[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      Xw[i] = d * (p0.R[i * 3 + 0] * hm[0] + p0.R[i * 3 + 1] * hm[1] + p0.R[i * 3 + 2] * hm[2]) + p0.t[i];
    float dX[3][6];
    dX_dT0(p1, Xw, dX);
    const float jx = -X[0] * inv_z * inv_z, jy = -X[1] * inv_z * inv_z;
#pragma unroll
    for (int j = 0; j < 6; ++j)
    {
      Q[0][j] = inv_z * dX[0][j] + jx * dX[2][j];
      Q[1][j] = inv_z * dX[1][j] + jy * dX[2][j];
      dXz[j] = dX[2][j];
    }
    Q[0][6] = rh[0] * inv_z - X[0] * rh[2] * inv_z * inv_z; // :324-325 without fx, fy
    Q[1][6] = rh[1] * inv_z - X[1] * rh[2] * inv_z * inv_z;
  }
  float S6[7], u6;
  {
    float GQ0[7], GQ1[7];
#pragma unroll
    for (int j = 0; j < 7; ++j)
    {
      // dead pixels: G and v are exactly zero (vm = 0), Q may hold inf/nan -> zero it so that every product vanishes
      Q[0][j] = live ? Q[0][j] : 0.f;
      Q[1][j] = live ? Q[1][j] : 0.f;
      GQ0[j] = G00 * Q[0][j] + G01 * Q[1][j];
      GQ1[j] = G01 * Q[0][j] + G11 * Q[1][j];
    }
#pragma unroll
    for (int i = 0; i < 7; ++i)
      S6[i] = Q[0][i] * GQ0[6] + Q[1][i] * GQ1[6];
    u6 = Q[0][6] * v0 + Q[1][6] * v1;
    sdd_acc += S6[6] * d * d; // (the photometric sigma only: the geometric edge keeps its own scale0-scale0 entry)
    float row8 = 0.f;
    if (MERGE)
    {
      // merged linearize: the geometric edge of the same pair (geometric_factor_kernels.cpp:671-716) at the same pixel --
      // {omega, D, grad D} from its kernel; a = dX_z/dT0 - gradD^T Jpi dX/dT0 and kappa = r_z - gradD^T dpi/dd in terms of
      // this kernel's own Q (Jpi's unit-focal form: P = diag(fx, fy) Q).  Its code0 column is kappa s0 b_n, so every block it
      // enters has the form of one the photometric contraction carries -- sigma b b^T, (c, sigma d, u6) b^T: the weights are
      // added and contracted once.  With wk = w_g omega kappa and g = (fx dD/dx, fy dD/dy):
      //   c_r  += wk a_r = wk dXz_r - Q0r (wk g_x) - Q1r (wk g_y)     (r < 6; the same form with dXz_6 := r_z gives sigma)
      //   u6   += wk rho,   row 8 = wk D  (the scale1-code0 block, read by the geometric finalize)
      // omega is zero for every pixel the geometric kernel found dead (the same pixels as here: same warp, same mask)
      const float om = in_range ? gp[0] : 0.f;
      const float gx = gp[2] * fx0, gy = gp[3] * fy0;
      const float kap = rh[2] - (gx * Q[0][6] + gy * Q[1][6]);
      const float wk = (prm.merge_w * om) * kap;
      const float wgx = wk * gx, wgy = wk * gy;
#pragma unroll
      for (int r = 0; r < 6; ++r)
        S6[r] += wk * dXz[r] - (Q[0][r] * wgx + Q[1][r] * wgy);
      S6[6] += wk * kap;
      u6 += wk * (gp[1] - X[2]);
      row8 = wk * gp[1];
    }
    // stash: the rows that multiply b_n (c (6), sigma*d, u6), sigma and the byte offset of the basis row, then the
    // operands of the pose tile
    f32x4 *st = reinterpret_cast<f32x4 *>(st_w + lane * kPhotoStashLD);
    st[0] = f32x4{S6[0], S6[1], S6[2], S6[3]};
    st[1] = f32x4{S6[4], S6[5], S6[6] * d, u6};
    st[2] = f32x4{S6[6], __int_as_float(my_loc * (CS * 4)), GQ0[0], GQ0[1]};
    st[3] = f32x4{GQ0[2], GQ0[3], GQ0[4], GQ0[5]};
    st[4] = f32x4{GQ1[0], GQ1[1], GQ1[2], GQ1[3]};
    st[5] = f32x4{GQ1[4], GQ1[5], v0, v1};
    st[6] = f32x4{Q[0][0], Q[0][1], Q[0][2], Q[0][3]};
    st[7] = f32x4{Q[0][4], Q[0][5], Q[1][0], Q[1][1]};
    st[8] = f32x4{Q[1][2], Q[1][3], Q[1][4], Q[1][5]};
    st[9] = f32x4{Q[0][6] * d, Q[1][6] * d, row8, 0.f};
  }
  __builtin_amdgcn_wave_barrier(); // same-wave LDS hand-over (in-order LDS pipe): no workgroup barrier needed
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

  // ---- MFMA contractions over this wave's 64 pixels, 4 pixels (K) per instruction; basis rows streamed from
  //      global memory in operand layout, AHEAD groups in flight ----
  SAGE_PHASE("D_contract");
  SAGE_TMARK(4);
  {
#if SAGE_PHOTO_ALT_ACC
    // extra accumulator sets of the three noise-critical tiles (live in this phase only): pixel group g goes to set g mod
    // (SAGE_PHOTO_ALT_ACC + 1), set 0 = acc
    f32x4 accb[SAGE_PHOTO_ALT_ACC][3];
#pragma unroll
    for (int u = 0; u < SAGE_PHOTO_ALT_ACC; ++u)
      accb[u][0] = accb[u][1] = accb[u][2] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
    const int i = lane & 15, k = lane >> 4;
    // rows 9..15 of the cross operand are zero: those lanes read slot 39 of the pixel's stash row (written as 0.f); row 8
    // (slot 38) is the merged linearize's scale1-code0 row, zero otherwise
    const int ai_slot = i < 8 ? i : (i == 8 ? 38 : 39);
    const uint32_t lane_off = (uint32_t)i * (NB == 2 ? 8u : 4u);
#ifndef SAGE_PHOTO_AHEAD
#define SAGE_PHOTO_AHEAD 6
#endif
    constexpr int G = 16, AHEAD = FS >= 32 ? SAGE_PHOTO_FS32_AHEAD : SAGE_PHOTO_AHEAD;
    float bl[G], bh[G], ai[G], sg[G], ya[G], yb[G];
    f32x2 vb[G]; // the loaded pairs stay whole until their wait (a half copied out earlier would be read before it landed)
    int locp[G];
#define SAGE_PHOTO_READ_STASH(g)                                                        \
  {                                                                                     \
    const float *p_ = st_w + ((g) * 4 + k) * kPhotoStashLD;                             \
    ai[g] = p_[ai_slot];                                                                \
    const f32x2 sl_ = *reinterpret_cast<const f32x2 *>(p_ + 8); /* sigma, loc */        \
    sg[g] = sl_[0];                                                                     \
    locp[g] = __float_as_int(sl_[1]);                                                   \
  }
#define SAGE_PHOTO_READ_POSE(g)                                                         \
  {                                                                                     \
    const float *p_ = st_w + ((g) * 4 + k) * kPhotoStashLD;                             \
    ya[g] = p_[10 + i];                                                                 \
    yb[g] = p_[24 + i];                                                                 \
  }
#define SAGE_PHOTO_ISSUE(g)                                                             \
  {                                                                                     \
    if (NB == 2)                                                                        \
      vb[g] = bload8(r_b0, (uint32_t)locp[g] + lane_off);                               \
    else                                                                                \
      vb[g][0] = bload4(r_b0, (uint32_t)locp[g] + lane_off);                            \
  }
#pragma unroll
    for (int g = 0; g < AHEAD + 1; ++g)
      SAGE_PHOTO_READ_STASH(g)
#pragma unroll
    for (int g = 0; g < AHEAD; ++g)
      SAGE_PHOTO_ISSUE(g)
    SAGE_PHOTO_READ_POSE(0)
    static_for<0, G>([&](auto gc) {
      constexpr int g = decltype(gc)::value; // (compile-time: the wait counts below are template arguments)
      if constexpr (g + 1 < G)
        SAGE_PHOTO_READ_POSE(g + 1) // LDS only: one group ahead is enough
      if constexpr (g + AHEAD < G)
        SAGE_PHOTO_ISSUE(g + AHEAD)
      if constexpr (g + AHEAD + 1 < G)
        SAGE_PHOTO_READ_STASH(g + AHEAD + 1)
      // (one load per group: younger than group g's are the groups up to g + AHEAD)
      vm_wait_keep2<(G - 1 - g < AHEAD ? G - 1 - g : AHEAD)>(vb[g]);
      bl[g] = vb[g][0];
      bh[g] = NB == 2 ? vb[g][1] : 0.f;
      const float a = ai[g];
#if SAGE_PHOTO_ALT_ACC
      if constexpr (CS == 32 && (g % (SAGE_PHOTO_ALT_ACC + 1)) != 0)
      {
        constexpr int NS = SAGE_PHOTO_ALT_ACC + 1;
        const int u = g % NS - 1;
        accb[u][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[g], yb[g], accb[u][2], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(sg[g] * bl[g], bl[g], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(sg[g] * bl[g], bh[g], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(sg[g] * bh[g], bh[g], acc[2], 0, 0, 0);
        accb[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bl[g], accb[u][0], 0, 0, 0);
        accb[u][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bh[g], accb[u][1], 0, 0, 0);
        return;
      }
#endif
      acc[YY] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[g], yb[g], acc[YY], 0, 0, 0);
      if constexpr (CS == 32)
      {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(sg[g] * bl[g], bl[g], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(sg[g] * bl[g], bh[g], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(sg[g] * bh[g], bh[g], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bl[g], acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bh[g], acc[4], 0, 0, 0);
      }
      else
      {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(sg[g] * bl[g], bl[g], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bl[g], acc[1], 0, 0, 0);
      }
    });
#if SAGE_PHOTO_ALT_ACC
    if (CS == 32)
    {
      // pairwise merge of the sets
#if SAGE_PHOTO_ALT_ACC == 3
      acc[3] = (acc[3] + accb[0][0]) + (accb[1][0] + accb[2][0]);
      acc[4] = (acc[4] + accb[0][1]) + (accb[1][1] + accb[2][1]);
      acc[YY] = (acc[YY] + accb[0][2]) + (accb[1][2] + accb[2][2]);
#else
#pragma unroll
      for (int u = 0; u < SAGE_PHOTO_ALT_ACC; ++u)
      {
        acc[3] += accb[u][0]; acc[4] += accb[u][1]; acc[YY] += accb[u][2];
      }
#endif
    }
#endif
  }
  __builtin_amdgcn_wave_barrier(); // the stash is rewritten by the next sub-tile
  SAGE_PHASE("E_second_level_flush");
  SAGE_TMARK(5);
  } // slice_live
#if SAGE_PHOTO_PREFETCH_A
  if (PACKED && sub + 1 < nsub)
  {
    pf_in = pf_in && (unsigned)pf_loc < (unsigned)(W0 * H0);
    pf_loc = pf_in ? pf_loc : 0;
    pf_d = pf_in ? E.dpt0[pf_loc] : 1.0f;
    pf_hm0 = pf_in ? pf_hm0 : 0.f; pf_hm1 = pf_in ? pf_hm1 : 0.f; pf_hm2 = pf_in ? pf_hm2 : 1.f;
  }
#endif
  // ---- second level: the LM step's distance from the exact step is set by the fp32 accumulation chains of the two
  //      cross tiles (rows c, sigma d, u6: the code gradient and the pose-code blocks) and of the pose tile; the code-code
  //      tiles do not matter (measured tile by tile, DESIGN s4).  After every sub-tile each lane moves its 12 values of
  //      those tiles into a wave-private LDS slot (fp32 add of 64-fmaf partial sums; no barrier, no record) and restarts
  //      their chains at zero; the last sub-tile of the run adds the slot back before the record is written. ----
  // (r06: written as three wave-uniform cases with the 12 LDS reads / writes of a case issued TOGETHER -- the per-element
  //  form compiled to 12 x [branch, ds_read, wait, add, branch, ds_write] in series, 1.7-2.4 k cycles per sub-tile in the wave
  //  timeline (profiles/r06_photo_wave_timeline.txt); `run_pos` counts the sub-tiles since the last record: no `sub % flush`)
  const bool last_of_run = sub + 1 == nsub || run_pos + 1 == flush;
  const bool first_of_run = run_pos == 0;
  if (nsub > 1 && !(first_of_run && last_of_run))
  {
    float *l2 = s_l2 + wave * (kPhotoL2Tiles * 256) + lane;
    constexpr int T0 = NT + 1 - kPhotoL2Tiles; // CS = 32: tiles 3, 4 (cross) and 5 (pose); CS = 16: every tile
    float prev[kPhotoL2Tiles][4];
    if (!first_of_run)
    {
#pragma unroll
      for (int t = 0; t < kPhotoL2Tiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          prev[t][r] = l2[(t * 4 + r) * 64];
    }
    else
    {
#pragma unroll
      for (int t = 0; t < kPhotoL2Tiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          prev[t][r] = 0.f;
    }
    if (last_of_run)
    {
#pragma unroll
      for (int t = 0; t < kPhotoL2Tiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[T0 + t][r] += prev[t][r];
    }
    else
    {
#pragma unroll
      for (int t = 0; t < kPhotoL2Tiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
          l2[(t * 4 + r) * 64] = prev[t][r] + acc[T0 + t][r];
          acc[T0 + t][r] = 0.f;
        }
    }
  }
  // ---- flush: one partial record per `flush` sub-tiles.  The fp32 accumulation chains (64 fmaf per sub-tile and
  //      accumulator) are what the LM step's distance from the exact step grows with (DESIGN s4); the workgroup keeps
  //      walking its run of sub-tiles (pose / descriptor prologue amortised, vertically adjacent bands stay in its L1/L2)
  const bool last_sub = sub + 1 == nsub;
  run_pos = last_of_run ? 0 : run_pos + 1;
  if (last_of_run)
  {
  // ---- cross-wave sum in a fixed order (deterministic): every wave dumps its tiles into its own slice of the (now idle)
  //      stash memory, one barrier, then all threads add the four slices as ((w0 + w1) + w2) + w3 on their way out to the
  //      partial record (a round of barriers per wave used to cost ~8 % of a one-sub-tile workgroup) ----
  // (every wave dumps into ITS OWN stash region -- it is done with it, the other waves' phase D is not disturbed: no
  //  barrier before the dump)
  constexpr int SLICE = 64 * kPhotoStashLD; // distance between the waves' regions
  static_assert(!JAC || (NT + 1) * 256 <= SLICE, "a wave's tiles must fit its own stash region");
#pragma unroll
  for (int t = 0; t < NT + 1; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      s_mem[wave * SLICE + t * 256 + r * 64 + lane] = acc[t][r];
  {
    const float se = wave_sum(err_acc), sn = wave_sum(vm_acc), sd = wave_sum(sdd_acc);
    if (lane == 63)
    {
      s_red[wave * 4 + 0] = sd;
      s_red[wave * 4 + 1] = se;
      s_red[wave * 4 + 2] = sn;
    }
  }
  __syncthreads();
  auto tsum = [&](int idx) { // element idx of the summed tiles
    return ((s_mem[idx] + s_mem[SLICE + idx]) + s_mem[2 * SLICE + idx]) + s_mem[3 * SLICE + idx];
  };
  auto tsumd = [&](int idx) { // the same in double (exact: four fp32 terms)
    return (((double)s_mem[idx] + (double)s_mem[SLICE + idx]) + (double)s_mem[2 * SLICE + idx]) + (double)s_mem[3 * SLICE + idx];
  };
  static_assert(kWaves == 4, "tsum adds four slices");
  constexpr int NCC = photo_cc_tiles(CS);
  float *out = prm.partials + (size_t)(rec_base + rec_idx) * photo_partial_floats(CS);
  ++rec_idx;
  double *outd = reinterpret_cast<double *>(out + photo_partial_double_offset(CS));
  if (tid < kPhotoScalars)
  {
    // scalar slots of the partial record (doubles): [0..20] Q^T G Q (upper triangle), [21..26] Q^T G q6 d,
    // [27] sigma d^2, [28..33] Q^T v, [34] q6^T v d, [35] error, [36] inliers.  Pose tile element (row r of A, col c of B):
    auto yy = [&](int r, int c) { return tsumd(YY * 256 + (r & 3) * 64 + ((r >> 2) * 16 + c)); };
    double a = 0.0;
    if (tid < 21)
    {
      int i = 0, rem = tid;
      while (rem >= 6 - i)
      {
        rem -= 6 - i;
        ++i;
      }
      const int j = i + rem; // sidx6(i, j) == tid, i <= j
      a = yy(j, i) + yy(6 + j, 6 + i);
    }
    else if (tid < 27)
      a = yy(tid - 21, 12) + yy(6 + tid - 21, 13);
    else if (tid >= 28 && tid < 34)
      a = yy(12, tid - 28) + yy(13, 6 + tid - 28);
    else if (tid == 34)
      a = yy(12, 12) + yy(13, 13);
    else if (tid == 27 || tid == 35 || tid == 36)
    {
      const int slot = tid == 27 ? 0 : tid - 34;
#pragma unroll
      for (int w = 0; w < kWaves; ++w)
        a += (double)s_red[w * 4 + slot];
    }
    outd[tid] = a;
  }
  for (int idx = tid; idx < NT * 256; idx += kBlock)
  {
    if (idx < NCC * 256) // code-code tiles: fp32
      out[kPhotoScalars + idx] = tsum(idx);
    else // cross tiles: double
      outd[kPhotoScalars + (idx - NCC * 256)] = tsumd(idx);
  }
    if (!last_sub)
    {
      __syncthreads(); // the slices are about to become stash memory again
#pragma unroll
      for (int t = 0; t < NT + 1; ++t)
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      err_acc = 0.f;
      vm_acc = 0.f;
      sdd_acc = 0.f;
    }
  }
  SAGE_PHASE("loop_end");
  SAGE_TMARK(6);
  } // sub-tile loop

  if (!JAC)
  {
    const float se = wave_sum(err_acc), sn = wave_sum(vm_acc), sg = wave_sum(gerr_acc);
    if (lane == 63)
    {
      s_red[wave * 4 + 0] = se;
      s_red[wave * 4 + 1] = sn;
      s_red[wave * 4 + 2] = sg;
      s_red[wave * 4 + 3] = sn; // the geometric edge counts the same pixels
    }
    __syncthreads();
    const int rec = prm.geo_loss_param > 0.f ? 4 : 2; // floats per workgroup record
    if (tid < rec)
    {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w)
        a += s_red[w * 4 + tid];
      prm.partials[(size_t)bid * rec + tid] = a;
    }
    return;
  }
}

// per-edge finalize (finalize_bodies.h), one workgroup per edge
template <int CS>
__global__ __launch_bounds__(kFinalizeBlock) void photo_finalize_kernel(const PhotoFinalizeParams prm)
{
  __shared__ double s[photo_finalize_lds_doubles(CS)];
  photo_finalize_body<CS>(prm, prm.edge_base + (int)blockIdx.x, s);
}

// error-only finalize: stats[e] = {sum(err)/n_in or 10*wsum, n_in}   (:1049-1058)
struct StatsFinalizeParams
{
  const int32_t *edge_first;
  const int32_t *edge_tiles;
  const float *partials; // [n_work][2]
  float *stats;
  float fallback; // 10*sum(w) or 10*weight
  float scale;    // 1 (photometric) or weight (geometric)
  int n_edges;
};

__global__ void stats_finalize_kernel(const StatsFinalizeParams prm)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= prm.n_edges)
    return;
  const int first = prm.edge_first[e], nt = prm.edge_tiles[e];
  float se = 0.f, sn = 0.f;
  for (int t = 0; t < nt; ++t)
  {
    se += prm.partials[(size_t)(first + t) * 2 + 0];
    sn += prm.partials[(size_t)(first + t) * 2 + 1];
  }
  prm.stats[2 * e + 0] = sn > 0.f ? prm.scale * se / sn : prm.fallback;
  prm.stats[2 * e + 1] = sn;
}

hipError_t launch_stats_finalize(hipStream_t s, const LaunchCommon &lc, float *stats, float fallback, float scale)
{
  StatsFinalizeParams fp{lc.edge_first, lc.edge_tiles, lc.partials, stats, fallback, scale, lc.n_edges};
  hipLaunchKernelGGL(stats_finalize_kernel, dim3((lc.n_edges + 63) / 64), dim3(64), 0, s, fp);
  return hipGetLastError();
}

static PhotoParams make_params(const PhotoEdge *single, const PhotoEdge *table, const LaunchCommon &lc,
                               const SagePyramid &pyr, const float *weights_host, float eps, float *wsum)
{
  PhotoParams p{};
  if (single)
    p.single = *single;
  p.table = table;
  p.work = lc.work;
  p.partials = lc.partials;
  p.pyr = pyr;
  float ws = 0.f;
  for (int l = 0; l < pyr.levels; ++l)
  {
    p.w[l] = weights_host[l];
    ws += weights_host[l];
  }
  p.eps = eps;
  p.tiles_per_block = lc.tiles_per_block;
  p.width = (int)pyr.cam[0].w;
  p.height = (int)pyr.cam[0].h;
  p.geo_loss_param = lc.fused_geo_loss_param;
  p.n_work = lc.n_work;
  p.rec_first = lc.flush > 0 ? lc.edge_first : nullptr;
  p.flush = lc.flush > 0 ? lc.flush : lc.tiles_per_block;
  p.merge_w = lc.merge_geo_weight;
  p.exact_coord = pyramid_is_dyadic(pyr) ? 0 : 1;
  for (int l = 0; l < pyr.levels; ++l)
  {
    p.rx[l] = pyr.cam[l].fx / pyr.cam[0].fx; // same fp32 quotient the kernels used to form per pixel
    p.ry[l] = pyr.cam[l].fy / pyr.cam[0].fy;
    p.lw[l] = (int)pyr.cam[l].w;
    p.lh[l] = (int)pyr.cam[l].h;
  }
  *wsum = ws;
  return p;
}

template <int CS, int FS>
static hipError_t photo_lin_impl(hipStream_t s, const PhotoEdge *single, const PhotoEdge *table,
                                 const LaunchCommon &lc, const SagePyramid &pyr, const float *wh, float eps,
                                 const EdgeOut &out)
{
  float wsum;
  PhotoParams p = make_params(single, table, lc, pyr, wh, eps, &wsum);
  if (lc.stage != 2)
  {
    if (lc.ev_start)
      (void)hipEventRecord(lc.ev_start, s);
    if (lc.packed && lc.merge_geo_weight > 0.f)
      hipLaunchKernelGGL((photo_kernel<CS, FS, true, 2>), dim3(lc.n_work), dim3(kBlock), 0, s, p);
    else if (lc.packed)
      hipLaunchKernelGGL((photo_kernel<CS, FS, true, 1>), dim3(lc.n_work), dim3(kBlock), 0, s, p);
    else
      hipLaunchKernelGGL((photo_kernel<CS, FS, true, 0>), dim3(lc.n_work), dim3(kBlock), 0, s, p);
    if (lc.ev_stop)
      (void)hipEventRecord(lc.ev_stop, s);
  }
  if (lc.stage == 1)
    return hipGetLastError();
  PhotoFinalizeParams f{};
  if (single)
    f.single = *single;
  f.table = table;
  f.edge_first = lc.edge_first;
  f.edge_tiles = lc.edge_tiles;
  f.partials = lc.partials;
  f.AtA = out.AtA;
  f.Atb = out.Atb;
  f.stats = out.stats;
  f.wide = out.wide;
  f.wsum = wsum;
  f.edge_base = lc.stage == 2 ? lc.edge_base : 0;
  const int n_fin = lc.stage == 2 ? lc.edge_count : lc.n_edges;
  if (n_fin > 0)
    hipLaunchKernelGGL((photo_finalize_kernel<CS>), dim3(n_fin), dim3(lc.fin_block), 0, s, f);
  return hipGetLastError();
}

template <int CS, int FS>
static hipError_t photo_err_impl(hipStream_t s, const PhotoEdge *single, const PhotoEdge *table,
                                 const LaunchCommon &lc, const SagePyramid &pyr, const float *wh, float eps,
                                 float *stats)
{
  float wsum;
  PhotoParams p = make_params(single, table, lc, pyr, wh, eps, &wsum);
  if (lc.ev_start)
    (void)hipEventRecord(lc.ev_start, s);
  if (lc.packed)
    hipLaunchKernelGGL((photo_kernel<CS, FS, false, 1>), dim3(lc.n_work), dim3(kBlock), 0, s, p);
  else
    hipLaunchKernelGGL((photo_kernel<CS, FS, false, 0>), dim3(lc.n_work), dim3(kBlock), 0, s, p);
  if (lc.ev_stop)
    (void)hipEventRecord(lc.ev_stop, s);
  if (lc.stage == 1) // the caller forms the per-edge statistics itself (window error pass)
    return hipGetLastError();
  return launch_stats_finalize(s, lc, stats, 10.0f * wsum, 1.0f);
}

hipError_t launch_photo_linearize(hipStream_t s, int CS, int FS, const PhotoEdge *single, const PhotoEdge *table,
                                  const LaunchCommon &lc, const SagePyramid &pyr, const float *weights_host,
                                  float eps, const EdgeOut &out)
{
  if (CS == 32 && FS == 16)
    return photo_lin_impl<32, 16>(s, single, table, lc, pyr, weights_host, eps, out);
  if (CS == 16 && FS == 16)
    return photo_lin_impl<16, 16>(s, single, table, lc, pyr, weights_host, eps, out);
  if (CS == 32 && FS == 32)
    return photo_lin_impl<32, 32>(s, single, table, lc, pyr, weights_host, eps, out);
  if (CS == 16 && FS == 32)
    return photo_lin_impl<16, 32>(s, single, table, lc, pyr, weights_host, eps, out);
  return hipErrorInvalidValue;
}

hipError_t launch_photo_error(hipStream_t s, int CS, int FS, const PhotoEdge *single, const PhotoEdge *table,
                              const LaunchCommon &lc, const SagePyramid &pyr, const float *weights_host,
                              float eps, float *stats)
{
  if (CS == 32 && FS == 16)
    return photo_err_impl<32, 16>(s, single, table, lc, pyr, weights_host, eps, stats);
  if (CS == 16 && FS == 16)
    return photo_err_impl<16, 16>(s, single, table, lc, pyr, weights_host, eps, stats);
  if (CS == 32 && FS == 32)
    return photo_err_impl<32, 32>(s, single, table, lc, pyr, weights_host, eps, stats);
  if (CS == 16 && FS == 32)
    return photo_err_impl<16, 32>(s, single, table, lc, pyr, weights_host, eps, stats);
  return hipErrorInvalidValue;
}

} // namespace sage

#ifdef SAGE_PHOTO_TRACE
extern "C" int sage_debug_photo_trace(void *dst, size_t bytes)
{
  const size_t all = sizeof(unsigned long long) * (size_t)sage::kTraceMaxWg * 4 * sage::kTraceSubs * sage::kTraceMarks;
  if (bytes > all)
    bytes = all;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sage::g_photo_trace), bytes, 0, hipMemcpyDeviceToHost);
}
extern "C" int sage_debug_photo_trace_clear()
{
  void *p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(sage::g_photo_trace)) != hipSuccess)
    return 1;
  return (int)hipMemset(p, 0, sizeof(unsigned long long) * (size_t)sage::kTraceMaxWg * 4 * sage::kTraceSubs * sage::kTraceMarks);
}
#endif
