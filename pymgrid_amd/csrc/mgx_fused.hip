// mgx_fused.hip -- the K-step kernels (step_k_kernel, rollout_kernel) of libmgx.so, one slice of the layouts per
// translation unit.  They are the bulk of the library's code (10 layouts x 16 specialisations each); compiled as
// MGX_FUSED_PARTS slices in parallel (-DMGX_FUSED_PART=p) they cost a fifth of the wall time of one translation unit:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -c -DMGX_FUSED_PART=p mgx_fused.hip -o mgx_fused_p.o
// The host side (mgx_abi.hip: mgx_step_k / mgx_rollout_discrete) offers a launch to every slice until one takes it.
#include "mgx_kernels.hpp"

#ifndef MGX_FUSED_PART
#error "compile with -DMGX_FUSED_PART=<0..MGX_FUSED_PARTS-1>"
#endif

// layouts (template parameter F) of each slice; together: 0..7, 14, 15 (MGX_DISPATCH_F in mgx_abi.hip)
#if MGX_FUSED_PART == 0
#define MGX_PART_FLAGS(X) X(0) X(1) X(2) X(4)
#elif MGX_FUSED_PART == 1
#define MGX_PART_FLAGS(X) X(3) X(5)
#elif MGX_FUSED_PART == 2
#define MGX_PART_FLAGS(X) X(6) X(7)
#elif MGX_FUSED_PART == 3
#define MGX_PART_FLAGS(X) X(14)
#elif MGX_FUSED_PART == 4
#define MGX_PART_FLAGS(X) X(15)
#elif MGX_FUSED_PART == 5
#define MGX_PART_FLAGS(X)
#else
#error "MGX_FUSED_PART out of range"
#endif

#define MGX_CAT2(a, b) a##b
#define MGX_CAT(a, b) MGX_CAT2(a, b)

#if MGX_FUSED_PART == 5
// The general path's K-step launch with the instance counts fixed at compile time (step_k_multi_small_kernel<F, CountsCT<...>>):
// (n_genset, n_battery, n_grid, n_load, n_pv) of the layouts that occur most; every other small layout takes the run-time-count
// form (CountsRT) of the same kernel.  The module list is the flags' (grid before battery: F_GRID_FIRST layouts are not listed).
namespace mgx {
#define MGX_STATIC_LAYOUTS(X) \
    X(7, 2, 2, 1, 1, 1) X(7, 2, 2, 2, 1, 1) X(7, 2, 2, 2, 2, 2) X(7, 1, 2, 1, 1, 1) X(7, 1, 2, 2, 1, 1) X(7, 2, 1, 1, 1, 1) X(7, 1, 1, 2, 1, 1) \
    X(3, 2, 2, 0, 1, 1) X(3, 2, 1, 0, 1, 1) X(3, 1, 2, 0, 1, 1) X(6, 0, 2, 1, 1, 1) X(6, 0, 2, 2, 1, 1) X(6, 0, 1, 2, 1, 1) \
    /* three modules of a kind (M = 3 instance slots; beyond the run-time-count register form, which holds two) */                       \
    X(7, 3, 3, 1, 1, 1) X(7, 3, 2, 1, 1, 1) X(7, 2, 3, 1, 1, 1) X(3, 3, 3, 0, 1, 1) X(6, 0, 3, 1, 1, 1)
bool launch_step_k_multi_static(const MultiStaticLaunch &L)
{
#define X(FV, G, B, R, LD, PV)                                                                                                           \
    if (L.flags == FV && L.ng == G && L.nb == B && L.nr == R && L.nl == LD && L.np == PV) {                                             \
        using CT = CountsCT<G, B, R, LD, PV>;                                                                                          \
        step_k_multi_small_kernel<FV, CT, CT::slots><<<L.blocks, BLOCK_MULTI, 0, L.stream>>>(*L.k, L.actions, L.t, L.K, L.normalized, L.out); \
        return true;                                                                                                                    \
    }
    MGX_STATIC_LAYOUTS(X)
#undef X
    return false;
}
bool launch_rollout_multi_static(const MultiStaticRollout &L)
{
#define X(FV, G, B, R, LD, PV)                                                                                                           \
    if (L.flags == FV && L.ng == G && L.nb == B && L.nr == R && L.nl == LD && L.np == PV) {                                             \
        using CT = CountsCT<G, B, R, LD, PV>;                                                                                          \
        rollout_multi_small_kernel<FV, CT, CT::slots><<<L.blocks, BLOCK_MULTI, 0, L.stream>>>(*L.k, L.lists, L.n_lists, L.list_len,     \
                                                                                             L.ids, L.per_step, L.t, L.K, L.out);      \
        return true;                                                                                                                    \
    }
    MGX_STATIC_LAYOUTS(X)
#undef X
    return false;
}
}  // namespace mgx
#else

namespace mgx {

// Register-ring depth of a specialisation: what a slot holds decides how deep the ring may be before it costs occupancy or spills --
// factorised series: the controls only (depth MGX_RING = 8; the write-bound full-output form MGX_RING_RICH = 16 without a GridModule);
// materialised series: + 2 series values per slot (8 for the hot form), + 6 with a GridModule (4, as before round 6)
template <int F, bool RC, bool FC>
constexpr int ring_depth()
{
    if (FC) return RC ? ((F & F_GRID) ? 8 : MGX_RING_RICH) : MGX_RING;
    return (F & F_GRID) ? 4 : (RC ? 4 : MGX_RING);
}

template <int F>
static void step_k_dispatch(const FusedLaunch &L)
{
#define MGX_STEP_K(AT, RC, FC) step_k_kernel<F, ring_depth<F, RC, FC>(), AT, RC, FC><<<L.blocks, BLOCK_K, 0, L.stream>>>( \
        *L.k, (const AT *)L.actions, L.t, L.K, L.normalized, L.out, L.gpb)
    if (L.act_f32) {
        if (L.fact) { if (L.rich) MGX_STEP_K(float, true, true); else MGX_STEP_K(float, false, true); }
        else { if (L.rich) MGX_STEP_K(float, true, false); else MGX_STEP_K(float, false, false); }
    } else {
        if (L.fact) { if (L.rich) MGX_STEP_K(double, true, true); else MGX_STEP_K(double, false, true); }
        else { if (L.rich) MGX_STEP_K(double, true, false); else MGX_STEP_K(double, false, false); }
    }
#undef MGX_STEP_K
}

template <int F>
static void rollout_dispatch(const FusedLaunch &L)
{
    // ring depth: a slot of a layout with a GridModule holds six values (depth 4), else two + an id byte (depth MGX_RING_ROLLOUT)
#define MGX_ROLLOUT(PS, RC, FC) rollout_kernel<F, (F & F_GRID) ? 4 : MGX_RING_ROLLOUT, PS, RC, FC><<<L.blocks, BLOCK_K, 0, L.stream>>>( \
        *L.k, *L.tab, L.ids, L.t, L.K, L.out, L.gpb)
    if (L.fact) {
        if (L.per_step) { if (L.rich) MGX_ROLLOUT(true, true, true); else MGX_ROLLOUT(true, false, true); }
        else { if (L.rich) MGX_ROLLOUT(false, true, true); else MGX_ROLLOUT(false, false, true); }
    } else {
        if (L.per_step) { if (L.rich) MGX_ROLLOUT(true, true, false); else MGX_ROLLOUT(true, false, false); }
        else { if (L.rich) MGX_ROLLOUT(false, true, false); else MGX_ROLLOUT(false, false, false); }
    }
#undef MGX_ROLLOUT
}

bool MGX_CAT(launch_step_k_p, MGX_FUSED_PART)(const FusedLaunch &L)
{
    switch (L.flags) {
#define X(FV) case FV: step_k_dispatch<FV>(L); return true;
        MGX_PART_FLAGS(X)
#undef X
        default: return false;
    }
}

bool MGX_CAT(launch_rollout_p, MGX_FUSED_PART)(const FusedLaunch &L)
{
    switch (L.flags) {
#define X(FV) case FV: rollout_dispatch<FV>(L); return true;
        MGX_PART_FLAGS(X)
#undef X
        default: return false;
    }
}

}  // namespace mgx
#endif
