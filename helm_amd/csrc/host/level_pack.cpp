// level_pack.cpp — re-times a levelised gate list into launches that fill the GPU.
//
// The reference walks the level map in order and fans each level out over the rayon pool
// (src/circuit.rs:524-543): a level is its unit of parallelism, and on a GPU every level ends
// with a partial round of workgroups.  Independent sub-circuits (the blocks of a batch, the
// S-boxes of one AES round) do not have to meet at level boundaries: a gate may run as soon as
// the gates that drive its inputs have run.  pack_levels() keeps exactly that dependency order
// and nothing else: list scheduling, lowest original level first, where a launch takes a whole
// number of `quantum` bootstraps (the engine's lockstep round, helm_hip_launch_quantum()) as
// long as that many are ready, and everything that is ready otherwise.  What a launch leaves
// behind stays ready and goes first in the next one.  Every gate still reads the ciphertexts the
// level schedule would have given it, so the outputs are bit-identical.
//
// Sequential circuits: a DFF (or any gate) that overwrites a wire an earlier-or-equal level still
// reads keeps its place after all packed launches, in its original level grouping; a netlist
// where such a gate also feeds later gates is left as it is (return value 1).
#include "helm_host.hpp"

#include <algorithm>
#include <functional>
#include <queue>

namespace helm {

static int pbs_cost(int op)
{
    switch (op) {
    case HELM_GATE_AND: case HELM_GATE_NAND: case HELM_GATE_OR: case HELM_GATE_NOR:
    case HELM_GATE_XOR: case HELM_GATE_XNOR: return 1;
    case HELM_GATE_MUX: return 2; // two bootstraps, one keyswitch (tfhe's mux)
    default: return 0;
    }
}

int pack_levels(const int32_t *op, const int32_t *in0, const int32_t *in1, const int32_t *in2, const int32_t *out,
                const int64_t *off, int64_t n_levels, int64_t quantum, std::vector<int64_t> &order,
                std::vector<int64_t> &new_off, const double *quarter_cost)
{
    const int64_t total = n_levels > 0 ? off[n_levels] : 0;
    order.clear();
    new_off.assign(1, 0);
    if (quantum < 1) throw Panic("pack_levels: quantum must be positive");
    auto keep = [&]() { // the level schedule, unchanged
        order.resize((size_t)total);
        for (int64_t g = 0; g < total; g++) order[(size_t)g] = g;
        new_off.assign(off, off + n_levels + 1);
        return 1;
    };
    if (total == 0) return keep();

    std::vector<int32_t> level((size_t)total);
    for (int64_t l = 0; l < n_levels; l++)
        for (int64_t g = off[l]; g < off[l + 1]; g++) level[(size_t)g] = (int32_t)l;
    int32_t max_wire = -1;
    for (int64_t g = 0; g < total; g++) {
        if (out[g] < 0) throw Panic("pack_levels: gate without an output wire");
        max_wire = std::max({max_wire, out[g], in0[g], in1[g], in2[g]});
    }
    std::vector<int64_t> writer((size_t)max_wire + 1, -1);
    for (int64_t g = 0; g < total; g++) {
        if (writer[(size_t)out[g]] >= 0) return keep(); // a wire driven twice: leave the schedule alone
        writer[(size_t)out[g]] = g;
    }
    // edges producer -> consumer for reads of a value produced at a LOWER level; a gate whose
    // output is read at its own or a lower level overwrites state (DFF) and goes to the tail
    std::vector<char> tail((size_t)total, 0);
    std::vector<int32_t> indeg((size_t)total, 0);
    std::vector<int64_t> head((size_t)total + 1, 0);
    auto each_input = [&](int64_t g, auto &&f) {
        const int32_t w[3] = {in0[g], in1[g], in2[g]};
        for (int q = 0; q < 3; q++) {
            if (w[q] < 0) continue;
            if (q == 1 && w[1] == w[0]) continue;
            if (q == 2 && (w[2] == w[0] || w[2] == w[1])) continue;
            f(w[q]);
        }
    };
    for (int64_t g = 0; g < total; g++)
        each_input(g, [&](int32_t w) {
            const int64_t x = writer[(size_t)w];
            if (x < 0) return;
            if (level[(size_t)x] < level[(size_t)g]) {
                head[(size_t)x + 1]++;
                indeg[(size_t)g]++;
            } else
                tail[(size_t)x] = 1;
        });
    for (int64_t g = 0; g < total; g++) head[(size_t)g + 1] += head[(size_t)g];
    std::vector<int64_t> succ((size_t)head[(size_t)total]), fill(head.begin(), head.end() - 1);
    for (int64_t g = 0; g < total; g++)
        each_input(g, [&](int32_t w) {
            const int64_t x = writer[(size_t)w];
            if (x >= 0 && level[(size_t)x] < level[(size_t)g]) succ[(size_t)fill[(size_t)x]++] = g;
        });
    for (int64_t g = 0; g < total; g++)
        if (tail[(size_t)g] && head[(size_t)g + 1] != head[(size_t)g]) return keep(); // state writer that also feeds later gates

    // list scheduling; the gate index is the priority (levels ascend with it).  Ready bootstrapped gates wait in
    // a min-heap, ready linear gates (no bootstrap: they cost nothing and unlock their consumers) all go at once:
    // O(total log total) whatever the quantum
    order.reserve((size_t)total);
    std::priority_queue<int64_t, std::vector<int64_t>, std::greater<int64_t>> heap;
    std::vector<int64_t> linear, skipped, launch;
    int64_t ready_pbs = 0;
    auto make_ready = [&](int64_t g) {
        const int c = pbs_cost(op[g]);
        if (c == 0) linear.push_back(g);
        else {
            heap.push(g);
            ready_pbs += c;
        }
    };
    for (int64_t g = 0; g < total; g++)
        if (!tail[(size_t)g] && indeg[(size_t)g] == 0) make_ready(g);
    // quarter_cost (optional): what a launch of at most 1/4, 2/4, 3/4, 4/4 of a round costs relative to a full round
    // (helm_hip_launch_costs(): the engine has a build per width - wide, duo, partial and full lockstep rounds).  A launch
    // narrower than a round then takes the width with the best bootstraps-per-cost among {everything that is ready, the
    // quarter steps below it} and leaves the rest for the next launch - as long as gates are still waiting for their
    // producers (at the drain nothing new will join the leftover: everything goes).  With 8 ranks a packed launch of a
    // 32-block AES batch is ~630 bootstraps per rank: 512 of them on the two-per-CU build (0.62 of a round's time) and 118
    // riding along with the next launch beat 630 in a 3/4-filled lockstep round (0.88).
    int64_t unreleased = 0; // bootstrapped gates not yet ready
    for (int64_t g = 0; g < total; g++)
        if (!tail[(size_t)g]) unreleased += pbs_cost(op[g]);
    auto pick_target = [&](int64_t ready, int64_t waiting) -> int64_t {
        if (ready >= quantum) return ready / quantum * quantum;
        // (quantum < 8: a quarter step could be ONE bootstrap, below the two a ready MUX costs - the launch would be empty)
        if (!quarter_cost || quantum < 8 || waiting < quantum) return ready;
        auto cost = [&](int64_t w) { return quarter_cost[std::min<int64_t>(3, (4 * w - 1) / quantum)]; };
        int64_t best = ready;
        double best_rate = (double)ready / cost(ready);
        for (int q = 3; q >= 1; q--) {
            const int64_t w = quantum * q / 4;
            if (w >= ready) continue;
            const double rate = (double)w / cost(w);
            if (rate > best_rate * 1.02) { // (a leftover costs launches later: only for a clear gain)
                best = w;
                best_rate = rate;
            }
        }
        return best;
    };
    while (!heap.empty() || !linear.empty()) {
        const int64_t target = pick_target(ready_pbs, unreleased - ready_pbs);
        int64_t taken = 0;
        launch.clear();
        skipped.clear();
        while (taken < target && !heap.empty() && skipped.size() < 64) { // (a bounded look-ahead past MUXes that do not fit)
            const int64_t g = heap.top();
            heap.pop();
            const int c = pbs_cost(op[g]);
            if (taken + c <= target) {
                taken += c;
                launch.push_back(g);
            } else
                skipped.push_back(g); // a two-bootstrap MUX that would overshoot: the next one-bootstrap gate fills the round
        }
        for (int64_t g : skipped) heap.push(g);
        ready_pbs -= taken;
        unreleased -= taken;
        launch.insert(launch.end(), linear.begin(), linear.end());
        linear.clear();
        if (launch.empty()) throw Panic("pack_levels: no progress"); // cannot happen: target >= one ready gate's cost
        std::sort(launch.begin(), launch.end());
        order.insert(order.end(), launch.begin(), launch.end());
        new_off.push_back((int64_t)order.size());
        for (int64_t g : launch)
            for (int64_t e = head[(size_t)g]; e < head[(size_t)g + 1]; e++)
                if (--indeg[(size_t)succ[(size_t)e]] == 0 && !tail[(size_t)succ[(size_t)e]]) make_ready(succ[(size_t)e]);
    }
    // state writers, in their original level grouping
    for (int64_t l = 0; l < n_levels; l++) {
        const size_t begin = order.size();
        for (int64_t g = off[l]; g < off[l + 1]; g++)
            if (tail[(size_t)g]) {
                if (indeg[(size_t)g] != 0) throw Panic("pack_levels: state writer with an unscheduled producer");
                order.push_back(g);
            }
        if (order.size() != begin) new_off.push_back((int64_t)order.size());
    }
    if ((int64_t)order.size() != total) throw Panic("pack_levels: combinational loop in a levelised netlist");
    return 0;
}

} // namespace helm
