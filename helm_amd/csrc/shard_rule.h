// shard_rule.h — how the gates of one launch are cut into `world` contiguous chunks (one per GPU).
//
// The sharded unit is the level of reference src/circuit.rs:531 (`gates.par_iter_mut()`); the reference has no
// multi-GPU path, rayon balances that loop dynamically.  Here the cut is static and made by BOOTSTRAP WEIGHT
// (binary gate = 1, MUX = 2, NOT / BUF / DFF / constants = 0), not by gate count: a launch whose
// free gates cluster at one end would otherwise give the ranks unequal work, and a sharded launch costs what its
// slowest rank costs.
//
// One definition, used by the engine (helm_hip.hip: shard tables, scatter table, gather-buffer slots) and exported
// through libhelm_host.so (helm_host_shard_bounds) so that hosts and the CPU tests cut exactly where the engine cuts.
#ifndef HELM_SHARD_RULE_H
#define HELM_SHARD_RULE_H

#include <cstdint>

namespace helm_shard {

// HELM_GATE_* of include/helm_hip.h: DFF 1, MUX 3, NOT 6, BUF 10, constants 11 / 12 cost no bootstrap.
inline int gate_weight(int op)
{
    switch (op) {
    case 3: return 2;                                   // MUX: two bootstraps, one keyswitch
    case 1: case 6: case 10: case 11: case 12: return 0; // DFF, NOT, BUF, CONST_ONE, CONST_ZERO
    default: return 1;
    }
}

// bounds[0 .. world]: rank r owns gates [bounds[r], bounds[r+1]) of the `count` gates with opcodes `op`.
// bounds[r] is the first gate g at which the bootstraps before g reach r/world of the launch's total, so every rank
// gets total/world bootstraps to within one gate (a MUX may put a rank one over); gates without a bootstrap travel with
// the rank whose range they fall into.  A launch without any bootstrap is cut by gate count.  Returns the largest chunk
// (= rows of one rank's slot in the all-gather, the same on every rank).
inline int64_t chunk_bounds(const int32_t *op, int64_t count, int world, int64_t *bounds)
{
    int64_t total = 0;
    for (int64_t g = 0; g < count; g++) total += gate_weight(op[g]);
    bounds[0] = 0;
    if (total == 0) {
        const int64_t chunk = (count + world - 1) / world;
        for (int r = 1; r <= world; r++) bounds[r] = chunk * r < count ? chunk * r : count;
    } else {
        int64_t g = 0, before = 0; // bootstraps of gates [0, g)
        for (int r = 1; r < world; r++) {
            while (g < count && before * world < total * r) before += gate_weight(op[g++]);
            bounds[r] = g;
        }
        bounds[world] = count;
    }
    int64_t rows = 0;
    for (int r = 0; r < world; r++)
        if (bounds[r + 1] - bounds[r] > rows) rows = bounds[r + 1] - bounds[r];
    return rows;
}

} // namespace helm_shard
#endif
