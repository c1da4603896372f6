// precision 'exact', block skipping: the POLICY -- what a launch decides from the measured costs of the handle's own earlier
// launches.  Pure functions of numbers (no HIP, no handle): the host code (exact_host.hpp) feeds them, som_policy_eval
// (include/somhip_test.h) exposes them to the CPU test suite (tests/test_policy_cpu.py).
//
// All costs are in milliseconds PER ROW of the launch they were measured on (so that launches of different sizes compare),
// except blk_ms (per 16-unit block run) and l2_ms_group (per (tile, group) pair level 1 kept).  0 = not measured yet.
#pragma once

namespace somhip {
namespace policy {

struct Costs {
    double full_total = 0.0;       // BMU search of the last launch WITHOUT a plan
    double full_screen = 0.0;      //   ... of which its screen
    double plan_total = 0.0;       // BMU search of the last launch under a plan
    double plan_over = 0.0;        //   ... less its screen (plan, lists, select, refine, re-score): launches without the scout
    double plan_over_scout = 0.0;  //   ... launches with the scout (+ nearest centroid, sort, gather, pick)
    double blk_ms = 0.0;           // screen ms per 16-unit block run under a plan
    double l2_ms_group = 0.0;      // level-2 ms per (tile, group) pair level 1 kept, when it last ran
    double l2_ratio = 1.0;         // blocks after level 2 / blocks after level 1, when it last ran
    double sort_ms = 0.0;          // sort + gather
};

constexpr int TILES_PER_GROUP = 4;   // 16-unit blocks of a 64-unit group

// the screen time per block a forecast works with: measured under a plan, else derived from a full scan (5 % for the list walk)
inline double block_ms(const Costs& c, double blocks_per_row) {
    return c.blk_ms > 0.0 ? c.blk_ms : c.full_screen > 0.0 ? 1.05 * c.full_screen / blocks_per_row : 0.0;
}
// what a scouted launch spends outside its screen: measured, else a full scan's own non-screen part + a fifth of its screen
inline double scouted_overhead(const Costs& c) {
    return c.plan_over_scout > 0.0 ? c.plan_over_scout : c.full_total > 0.0 ? (c.full_total - c.full_screen) + 0.2 * c.full_screen : 0.0;
}

// COMMIT a scouted plan whose sample tiles forecast `share` of the blocks?  Priced: share x blocks x block time + overhead must
// stay 3 % under the last launch without a plan.  Not priced yet (no full scan on record): declined above 0.8 of the blocks.
inline bool commit_scouted_plan(const Costs& c, double share, double blocks_per_row) {
    const double blk = block_ms(c, blocks_per_row), over = scouted_overhead(c);
    if (blk > 0.0 && over > 0.0 && c.full_total > 0.0) return share * blocks_per_row * blk + over < 0.97 * c.full_total;
    return !(share > 0.8);
}

// LEVEL 2 from a sample that ran both levels (share after level 2, after level 1): it removes (1 - ratio) of a kept group's
// four blocks at its measured -- else: a sixth of the group's screen -- cost per kept group.  Nothing priced: below a ratio of 0.85.
inline bool level2_from_sample(const Costs& c, double share, double share1, double blocks_per_row) {
    if (!(share1 > 0.0)) return true;
    const double ratio = share / share1, blk = block_ms(c, blocks_per_row);
    const double l2c = c.l2_ms_group > 0.0 ? c.l2_ms_group : blk > 0.0 ? TILES_PER_GROUP * blk / 6.0 : 0.0;
    if (l2c > 0.0 && blk > 0.0) return (1.0 - ratio) * TILES_PER_GROUP * blk > l2c;
    return ratio < 0.85;
}
// ... and from the launch that just ran it (both measured); before that: round 4's fitted rule on the two shares
inline bool level2_pays(const Costs& c, double share, double share1) {
    if (c.l2_ms_group > 0.0 && c.blk_ms > 0.0) return (1.0 - c.l2_ratio) * TILES_PER_GROUP * c.blk_ms > c.l2_ms_group;
    return 1.5 * (share1 - share) > 0.1 * share1 + 0.006;
}

// did a SORT pay?  The blocks it saved against the stale order's share, at the measured screen time per block, over the epochs
// the order will serve, against the measured sort + gather (before those are measured: the share fell by 7 % or more)
inline bool sort_paid(const Costs& c, double share_stale, double share_fresh, double blocks_per_row, int epochs_served) {
    if (c.blk_ms > 0.0 && c.sort_ms > 0.0) return (share_stale - share_fresh) * blocks_per_row * c.blk_ms * (double)epochs_served > c.sort_ms;
    return share_fresh <= 0.93 * share_stale;
}

// an IDLE plan: it ran more than half of the blocks and cost what the last launch without a plan cost (with no such launch on
// record: it kept more than 0.97 of the blocks).  With most blocks proven empty a slow launch is somebody else's kernels on the card.
inline bool plan_idle(const Costs& c, double share) {
    return share > 0.5 && (c.full_total > 0.0 ? c.plan_total >= 0.97 * c.full_total : share > 0.97);
}

// does the scout GO ON beside last epoch's BMUs?  Its picks beat them by a tenth of the squared distance on a quarter of the rows
// AND halving the screen would still pay for it: (last share) x (screen time per block) / 2 against what a scouted launch spends
// beyond an unscouted one outside its screen (before that is measured: a tenth of a full screen)
inline bool scout_continues(const Costs& c, double win_share, double share_last, double blocks_per_row) {
    if (!(win_share >= 0.25)) return false;
    const double blk = c.blk_ms > 0.0 ? c.blk_ms : c.full_screen > 0.0 ? c.full_screen / blocks_per_row : 0.0;
    const double sc = (c.plan_over_scout > 0.0 && c.plan_over > 0.0) ? c.plan_over_scout - c.plan_over : 0.1 * c.full_screen;
    return blk > 0.0 && sc > 0.0 && 0.5 * share_last * blocks_per_row * blk > sc;
}

// is a row set large enough for the scout's fixed part (some thirty small launches, a quarter of a millisecond) to pay?
inline bool rows_worth_a_scout(double n_rows, double units, double features) { return n_rows * units * features >= 3.0e11; }

}  // namespace policy
}  // namespace somhip
