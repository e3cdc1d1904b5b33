// gate_circuit.cpp — EncWireMap and GateCircuit: the encrypted evaluator of gates mode.
// Mirrors reference src/circuit.rs:449-577 (impl EvalCircuit<CtxtBool> for GateCircuit);
// the per-gate ServerKey calls of gates.rs:254-275 become one helm_hip level per
// netlist level.
#include "helm_host.hpp"
#include <unordered_map>

#include <algorithm>
#include <atomic>
#include <sstream>

namespace helm {

uint64_t next_map_id()
{
    static std::atomic<uint64_t> next{1};
    return next.fetch_add(1);
}

static void hip_ok(int rc, const char *what)
{
    if (rc != 0) throw Panic(std::string(what) + ": " + helm_hip_last_error());
}
static void client_ok(int rc, const char *what)
{
    if (rc != 0) throw Panic(std::string(what) + ": " + helm_client_last_error());
}

// ---------------------------------------------------------------------------------------
// EncWireMap
// ---------------------------------------------------------------------------------------
EncWireMap::~EncWireMap()
{
    if (wires_) helm_hip_wires_free(ctx_, wires_);
}

std::vector<std::string> EncWireMap::keys() const
{
    std::vector<std::string> k;
    k.reserve(index_.size());
    for (auto &kv : index_) k.push_back(kv.first);
    std::sort(k.begin(), k.end());
    return k;
}

int EncWireMap::row(const std::string &k) const
{
    auto it = index_.find(k);
    if (it == index_.end()) throw Panic("wire \"" + k + "\" not in the encrypted wire map");
    return it->second;
}

static void copy_rows(helm_hip_ctx *ctx, helm_hip_wires *src, helm_hip_wires *dst, int64_t used, int /*n*/)
{
    if (used <= 0) return;
    std::vector<int32_t> idx((size_t)used);
    for (int64_t i = 0; i < used; i++) idx[(size_t)i] = (int32_t)i;
    hip_ok(helm_hip_wires_copy(ctx, src, idx.data(), dst, idx.data(), used), "wires_copy"); // on the device
}

void EncWireMap::grow(int64_t rows)
{
    if (rows <= cap_) return;
    const int64_t want = std::max<int64_t>(rows, std::max<int64_t>(16, cap_ * 2));
    helm_hip_wires *nw = nullptr;
    hip_ok(helm_hip_wires_alloc(ctx_, want, &nw), "wires_alloc");
    if (wires_) {
        copy_rows(ctx_, wires_, nw, std::min<int64_t>(cap_, (int64_t)index_.size()), n_);
        helm_hip_wires_free(ctx_, wires_);
    }
    wires_ = nw;
    cap_ = want;
}

int EncWireMap::scratch(int64_t rows)
{
    grow((int64_t)index_.size() + rows);
    return (int)index_.size();
}

void EncWireMap::reserve_keys(const std::vector<std::string> &names)
{
    const int64_t before = (int64_t)index_.size();
    // rows already holding data must be preserved by grow(): extend the index afterwards
    std::vector<const std::string *> fresh;
    std::unordered_map<std::string, int> seen;
    for (auto &nm : names)
        if (!index_.count(nm) && !seen.count(nm)) {
            seen[nm] = 1;
            fresh.push_back(&nm);
        }
    grow(before + (int64_t)fresh.size());
    for (auto *nm : fresh) {
        const int r = (int)index_.size();
        index_[*nm] = r;
    }
    gen_++;
}

std::vector<uint32_t> EncWireMap::get(const std::string &k) const
{
    const int32_t r = row(k);
    std::vector<uint32_t> out((size_t)n_ + 1);
    hip_ok(helm_hip_wires_download(ctx_, wires_, &r, out.data(), 1), "wires_download");
    return out;
}

void EncWireMap::insert(const std::string &k, const uint32_t *lwe)
{
    auto it = index_.find(k);
    int32_t r;
    if (it == index_.end()) {
        r = (int32_t)index_.size();
        grow(r + 1);
        index_[k] = r;
    } else
        r = it->second;
    hip_ok(helm_hip_wires_upload(ctx_, wires_, &r, lwe, 1), "wires_upload");
    gen_++;
}

std::unique_ptr<EncWireMap> EncWireMap::clone() const
{
    auto m = std::make_unique<EncWireMap>(ctx_, n_);
    const int64_t used = (int64_t)index_.size();
    if (used > 0) {
        m->grow(used);
        copy_rows(ctx_, wires_, m->wires_, used, n_);
    }
    m->index_ = index_;
    return m;
}

// ---------------------------------------------------------------------------------------
// GateCircuit
// ---------------------------------------------------------------------------------------
GateCircuit::GateCircuit(helm_client_key *client_key, helm_hip_ctx *server_key, Circuit circuit)
    : client_key_(client_key), server_key_(server_key), circuit_(std::move(circuit))
{
    helm_hip_params P;
    client_ok(helm_client_params(client_key, &P), "client_params");
    n_ = P.n;
}

GateCircuit::~GateCircuit()
{
    if (prog_) helm_hip_program_destroy(server_key_, prog_);
}

void GateCircuit::shard_over(helm_comm *comm, int64_t replicate_below)
{
    int world = 1;
    if (comm) hip_ok(helm_comm_info(comm, nullptr, &world, nullptr, nullptr), "comm_info");
    if (replicate_below < 0) throw Panic("shard_over: replicate_below must not be negative");
    if (comm != comm_ || world != comm_world_) { // the launches are packed for the world size: plan the program again
        if (prog_) helm_hip_program_destroy(server_key_, prog_);
        prog_ = nullptr;
    }
    comm_ = comm;
    comm_world_ = world;
    replicate_below_ = replicate_below;
    memo_.valid = false;
}

// reference src/circuit.rs:450-480
std::unique_ptr<EncWireMap> GateCircuit::encrypt_inputs(const std::set<std::string> &wire_set,
                                                        const std::map<std::string, PtxtType> &input_wire_map)
{
    auto m = std::make_unique<EncWireMap>(server_key_, n_);
    std::vector<std::string> names(wire_set.begin(), wire_set.end());
    names.insert(names.end(), circuit_.input_wires().begin(), circuit_.input_wires().end());
    names.insert(names.end(), circuit_.dff_outputs().begin(), circuit_.dff_outputs().end());
    m->reserve_keys(names);
    // every gate-output wire starts as server_key.trivial_encrypt(false)
    {
        std::vector<int32_t> idx;
        for (auto &w : wire_set) idx.push_back(m->row(w));
        std::vector<uint8_t> zeros(idx.size(), 0);
        if (!idx.empty())
            hip_ok(helm_hip_wires_set_trivial(server_key_, m->table(), idx.data(), zeros.data(), (int64_t)idx.size()),
                   "wires_set_trivial");
    }
    std::vector<int32_t> idx;
    std::vector<uint8_t> bits;
    const bool dummy = input_wire_map.empty() || input_wire_map.count("dummy");
    for (auto &input_wire : circuit_.input_wires()) {
        bool v = false;
        if (!dummy) {
            auto it = input_wire_map.find(input_wire);
            if (it == input_wire_map.end()) throw Panic("\n Input wire \"" + input_wire + "\" not in input wires!");
            if (it->second.kind != PtxtType::Bool) throw Panic("internal error: entered unreachable code");
            v = it->second.as_bool();
        }
        idx.push_back(m->row(input_wire));
        bits.push_back(v ? 1 : 0);
    }
    // DFF outputs are also input wires (verilog_parser.rs:225-227); the reference inserts them a
    // second time with encrypt(false), which wins (circuit.rs:474-476).  One upload must not carry
    // the same row twice (rows are written concurrently): override in place.
    for (auto &w : circuit_.dff_outputs()) {
        const int32_t r = m->row(w);
        size_t q = 0;
        while (q < idx.size() && idx[q] != r) q++;
        if (q < idx.size()) bits[q] = 0;
        else {
            idx.push_back(r);
            bits.push_back(0);
        }
    }
    if (!idx.empty()) {
        std::vector<uint32_t> cts(idx.size() * (size_t)(n_ + 1));
        client_ok(helm_client_encrypt_bool(client_key_, bits.data(), (int64_t)bits.size(), cts.data()), "encrypt");
        hip_ok(helm_hip_wires_upload(server_key_, m->table(), idx.data(), cts.data(), (int64_t)idx.size()), "wires_upload");
    }
    return m;
}

// reference src/circuit.rs:482-490
std::unique_ptr<EncWireMap> GateCircuit::init_ready()
{
    auto m = std::make_unique<EncWireMap>(server_key_, n_);
    m->reserve_keys(circuit_.output_wires());
    std::vector<int32_t> idx;
    for (auto &w : circuit_.output_wires()) idx.push_back(m->row(w));
    std::vector<uint8_t> zeros(idx.size(), 0);
    if (!idx.empty())
        hip_ok(helm_hip_wires_set_trivial(server_key_, m->table(), idx.data(), zeros.data(), (int64_t)idx.size()),
               "wires_set_trivial");
    return m;
}

// reference src/circuit.rs:492-504: valid = mux(READY, enc_value, valid) per output.
void GateCircuit::evaluate_ready(const EncWireMap &enc_wire_map, EncWireMap &valid_outputs)
{
    std::vector<std::string> keys;
    for (auto &k : valid_outputs.keys())
        if (enc_wire_map.contains_key(k)) keys.push_back(k);
    if (keys.empty()) return;
    if (!enc_wire_map.contains_key("READY")) throw Panic("called `Option::unwrap()` on a `None` value (READY)");
    // one MUX level over a scratch table: row 0 = READY, then (then, else) pairs
    EncWireMap tmp(server_key_, n_);
    std::vector<std::string> names = {"READY"};
    for (auto &k : keys) {
        names.push_back("t:" + k);
        names.push_back("e:" + k);
    }
    tmp.reserve_keys(names);
    tmp.insert("READY", enc_wire_map.get("READY").data());
    std::vector<int32_t> op, i0, i1, i2, out;
    for (auto &k : keys) {
        tmp.insert("t:" + k, enc_wire_map.get(k).data());
        tmp.insert("e:" + k, valid_outputs.get(k).data());
        op.push_back(HELM_GATE_MUX);
        i0.push_back(tmp.row("t:" + k));
        i1.push_back(tmp.row("e:" + k));
        i2.push_back(tmp.row("READY"));
        out.push_back(tmp.row("e:" + k));
    }
    hip_ok(helm_hip_eval_gate_level(server_key_, tmp.table(), op.data(), i0.data(), i1.data(), i2.data(), out.data(),
                                    (int64_t)op.size()),
           "eval_gate_level");
    for (auto &k : keys) valid_outputs.insert(k, tmp.get("e:" + k).data());
}

// reference src/circuit.rs:506-549
std::unique_ptr<EncWireMap> GateCircuit::evaluate_encrypted(const EncWireMap &enc_wire_map, size_t cycle,
                                                            const std::string & /*ptxt_type*/)
{
    if (!circuit_.gates_empty()) throw Panic("assertion failed: self.circuit.gates.is_empty()");
    if (!circuit_.get_ordered_gates().empty()) throw Panic("assertion failed: self.circuit.ordered_gates.is_empty()");
    // same-cycle memo (gates.rs:55-59): this cycle was already evaluated on this very map -> no launch
    if (memo_.hit(cycle, enc_wire_map)) {
        memo_hits_++;
        log_ += "  Cycle " + std::to_string(cycle) + " already evaluated on these inputs: cached wire map returned\n";
        return memo_.out->clone();
    }
    auto eval_values = enc_wire_map.clone();

    // (re)build the device program when the name -> row layout changed
    const std::vector<std::string> keys = eval_values->keys();
    bool same = prog_ != nullptr && keys.size() == prog_keys_.size();
    if (same)
        for (size_t i = 0; i < keys.size() && same; i++)
            same = keys[i] == prog_keys_[i] && eval_values->row(keys[i]) == prog_rows_[i];
    if (!same) {
        if (prog_) helm_hip_program_destroy(server_key_, prog_);
        prog_ = nullptr;
        std::vector<int32_t> op, i0, i1, i2, out;
        std::vector<int64_t> off = {0};
        level_end_.clear();
        n_scratch_ = 0;
        const int32_t scratch_base = (int32_t)keys.size(); // scratch rows sit behind the named rows
        for (auto &kv : circuit_.level_map()) {
            // compute_levels (circuit.rs:174-239) keeps the combinational gates of a level independent, but every DFF goes
            // to ONE last level, and a flip-flop fed by another (`dff g1(d, q1); dff g2(q1, q2);`) reads a wire written in
            // that same level.  The reference evaluates such a level with par_iter (circuit.rs:531, also in its plaintext
            // evaluator, :348-381): whichever gate comes first.  This repository's plaintext evaluator gives the level
            // snapshot semantics - every gate reads the values from before the level, as flip-flops on one clock edge do -
            // and the encrypted evaluation follows it: rows that one gate of the level writes and ANOTHER reads are copied
            // to scratch rows in a level of BUF gates put in front, and the readers read the copies (the engine refuses a
            // level with a read-after-write inside; a gate that rewrites its own operand needs no copy).
            const size_t first = op.size();
            std::unordered_map<int32_t, size_t> writer; // row -> gate (index into op) that writes it in this level
            for (auto &gate : kv.second) {
                const auto &ins = gate.get_input_wires();
                auto in_row = [&](size_t i) -> int32_t { return i < ins.size() ? eval_values->row(ins[i]) : -1; };
                const GateType t = gate.get_gate_type();
                if ((t == GateType::Mux && ins.size() < 3) ||
                    ((t == GateType::And || t == GateType::Nand || t == GateType::Or || t == GateType::Nor ||
                      t == GateType::Xor || t == GateType::Xnor) && ins.size() < 2) ||
                    ((t == GateType::Not || t == GateType::Buf || t == GateType::Dff) && ins.empty()))
                    throw Panic("index out of bounds: gate " + gate.get_gate_name() + " has too few inputs");
                writer.emplace(eval_values->row(gate.get_output_wire()), op.size()); // (two writers: the engine refuses the level)
                op.push_back((int32_t)t);
                i0.push_back(in_row(0));
                i1.push_back(in_row(1));
                i2.push_back(in_row(2));
                out.push_back(eval_values->row(gate.get_output_wire()));
            }
            std::map<int32_t, int32_t> copy_of; // hazard row -> scratch row
            for (size_t g = first; g < op.size(); g++)
                for (std::vector<int32_t> *in : {&i0, &i1, &i2}) {
                    int32_t &r = (*in)[g];
                    auto it = r >= 0 ? writer.find(r) : writer.end();
                    if (it == writer.end() || it->second == g) continue;
                    auto c = copy_of.find(r);
                    if (c == copy_of.end()) c = copy_of.emplace(r, scratch_base + (int32_t)n_scratch_++).first;
                    r = c->second;
                }
            if (!copy_of.empty()) { // the copies form a level of their own in front of the level
                const size_t n = copy_of.size();
                for (std::vector<int32_t> *v : {&op, &i0, &i1, &i2, &out}) v->insert(v->begin() + (long)first, n, -1);
                size_t q = first;
                for (auto &c : copy_of) {
                    op[q] = (int32_t)GateType::Buf;
                    i0[q] = c.first;
                    out[q] = c.second;
                    q++;
                }
                off.push_back((int64_t)(first + n));
            }
            off.push_back((int64_t)op.size());
            level_end_.push_back((int64_t)off.size() - 1);
        }
        // launch packing (level_pack.cpp): wide levels are re-timed into whole lockstep rounds; a netlist
        // whose levels are narrower than one round keeps its level schedule unchanged
        {
            const int64_t quantum = helm_hip_launch_quantum(server_key_) * comm_world_; // a round per rank
            double quarter_cost[4]; // the engine's cost per launch width: launches narrower than a round take its best width
            const bool costed = helm_hip_launch_costs(server_key_, quarter_cost) == 0;
            std::vector<int64_t> order, poff;
            if (quantum > 0 &&
                pack_levels(op.data(), i0.data(), i1.data(), i2.data(), out.data(), off.data(), (int64_t)off.size() - 1,
                            quantum, order, poff, costed ? quarter_cost : nullptr) == 0 &&
                poff != off) {
                auto permute = [&](std::vector<int32_t> &v) {
                    std::vector<int32_t> t(v.size());
                    for (size_t q = 0; q < order.size(); q++) t[q] = v[(size_t)order[q]];
                    v.swap(t);
                };
                permute(op); permute(i0); permute(i1); permute(i2); permute(out);
                off = poff;
                packed_ = true;
            } else
                packed_ = false;
        }
        if (int rc = helm_hip_program_create(server_key_, op.data(), i0.data(), i1.data(), i2.data(), out.data(),
                                             off.data(), (int64_t)off.size() - 1, &prog_)) {
            (void)rc;
            // same messages as gates.rs:257-264 for LUT / arithmetic gates in boolean mode
            throw Panic(std::string("program_create: ") + helm_hip_last_error());
        }
        prog_keys_ = keys;
        prog_rows_.clear();
        for (auto &k : keys) prog_rows_.push_back(eval_values->row(k));
        prog_launches_ = (int64_t)off.size() - 1;
        pbs_count_ = 0;
        for (int64_t l = 0; l < prog_launches_; l++) pbs_count_ += helm_hip_program_level_pbs(prog_, l);
    }
    if (n_scratch_ > 0) eval_values->scratch(n_scratch_); // the copies of rows a level both reads and rewrites
    const int64_t total_levels = (int64_t)circuit_.level_map().size();
    std::ostringstream os;
    if (comm_) {
        // every launch of more than replicate_below_ bootstraps split over the ranks, outputs all-gathered inside the engine
        hip_ok(helm_hip_program_run_sharded_comm(server_key_, prog_, eval_values->table(), comm_, replicate_below_, overlap_ ? 1 : 0), "program_run_sharded_comm");
        os << "  Evaluated gates of " << total_levels << " levels in " << prog_launches_ << " launches sharded over " << comm_world_
           << " rank(s)\n";
    } else if (!packed_) {
        int64_t l = 0;
        size_t li = 0;
        for (auto &kv : circuit_.level_map()) {
            const int64_t end = level_end_[li++]; // a level of chained flip-flops spans several program levels
            hip_ok(helm_hip_program_run(server_key_, prog_, eval_values->table(), l, end), "program_run");
            os << "  Evaluated gates in level [" << kv.first << "/" << total_levels << "]\n";
            l = end;
        }
    } else {
        hip_ok(helm_hip_program_run(server_key_, prog_, eval_values->table(), 0, prog_launches_), "program_run");
        os << "  Evaluated gates of " << total_levels << " levels in " << prog_launches_ << " packed launches\n";
    }
    hip_ok(helm_hip_sync(server_key_), "sync");
    log_ += os.str();
    memo_.out = eval_values->clone();
    memo_.cycle = cycle;
    memo_.in_id = enc_wire_map.id();
    memo_.in_gen = enc_wire_map.generation();
    memo_.valid = true;
    return eval_values;
}

// reference src/circuit.rs:551-576
std::map<std::string, PtxtType> GateCircuit::decrypt_outputs(const EncWireMap &enc_wire_map, bool verbose)
{
    std::map<std::string, PtxtType> decrypted_outputs;
    for (auto &output_wire : circuit_.output_wires()) {
        const std::vector<uint32_t> ct = enc_wire_map.get(output_wire);
        uint8_t bit = 0;
        client_ok(helm_client_decrypt_bool(client_key_, ct.data(), 1, &bit), "decrypt");
        decrypted_outputs[output_wire] = PtxtType::boolean(bit != 0);
    }
    std::ostringstream os;
    size_t i = 0;
    for (auto &kv : decrypted_outputs) {
        if (i > 10 && !verbose) {
            os << "[!] More than ten output_wires, pass `--verbose` to see output.\n";
            break;
        }
        os << " " << kv.first << ": " << kv.second.to_string() << "\n";
        i++;
    }
    log_ += os.str();
    return decrypted_outputs;
}

} // namespace helm
