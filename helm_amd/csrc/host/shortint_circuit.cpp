// shortint_circuit.cpp — SiEncWireMap, RadixEngine, LutCircuit and ArithCircuit: the
// encrypted evaluators of LUT mode and arithmetic mode over include/helm_shortint.h.
//
//   LutCircuit    reference src/circuit.rs:75-79, 393-401, 969-1120 (impl EvalCircuit<CtxtShortInt>)
//                 + gates::lut(), src/gates.rs:754-785 -> helm_si_eval_lut_level per level
//   ArithCircuit  reference src/circuit.rs:81-85, 403-411, 1112-1500 (impl EvalCircuit<FheType>)
//                 + Gate::evaluate_encrypted_{add,sub,mul,copy}_block(_plain),
//                 src/gates.rs:306-702 -> RadixEngine (FheUint8..128 = radix integers of
//                 2-bit-message shortint blocks, least significant block first)
//
// The radix algorithms inside tfhe's FheUintN operators live in the absent `tfhe` crate;
// the ones here are level-batched restatements with the same results mod 2^bits (what the
// reference's tests pin: tests/gates_test.rs:127-310, tests/circuit_test.rs:349-368).
#include "helm_host.hpp"
#include <functional>
#include <unordered_map>
#include <unordered_set>
#include <thread>

#include <algorithm>
#include <chrono>
#include <sstream>

namespace helm {

static void si_ok(int rc, const char *what)
{
    if (rc != 0) throw Panic(std::string(what) + ": " + helm_hip_last_error());
}

// ---------------------------------------------------------------------------------------
// SiEncWireMap
// ---------------------------------------------------------------------------------------
SiEncWireMap::SiEncWireMap(helm_si_ctx *ctx, int blocks) : ctx_(ctx), blocks_(blocks)
{
    helm_si_params P;
    si_ok(helm_si_get_params(ctx, &P), "get_params");
    dim_ = P.k * P.N;
}

SiEncWireMap::~SiEncWireMap()
{
    if (wires_) helm_si_wires_free(ctx_, wires_);
}

std::vector<std::string> SiEncWireMap::keys() const
{
    std::vector<std::string> k;
    k.reserve(index_.size());
    for (auto &kv : index_) k.push_back(kv.first);
    std::sort(k.begin(), k.end());
    return k;
}

int SiEncWireMap::row(const std::string &k) const
{
    auto it = index_.find(k);
    if (it == index_.end()) throw Panic("wire \"" + k + "\" not in the encrypted wire map");
    return it->second;
}

void SiEncWireMap::grow(int64_t rows)
{
    if (rows <= cap_) return;
    const int64_t want = std::max<int64_t>(rows, std::max<int64_t>(16, cap_ * 2));
    helm_si_wires *nw = nullptr;
    si_ok(helm_si_wires_alloc(ctx_, want, &nw), "wires_alloc");
    if (wires_) {
        const int64_t used = std::min<int64_t>(cap_, (int64_t)index_.size() * blocks_);
        if (used > 0) {
            std::vector<int32_t> idx((size_t)used);
            for (int64_t i = 0; i < used; i++) idx[(size_t)i] = (int32_t)i;
            si_ok(helm_si_wires_copy(ctx_, wires_, idx.data(), nw, idx.data(), used), "wires_copy");
        }
        helm_si_wires_free(ctx_, wires_);
    }
    wires_ = nw;
    cap_ = want;
}

void SiEncWireMap::reserve_keys(const std::vector<std::string> &names, int64_t scratch_rows)
{
    std::vector<const std::string *> fresh;
    std::unordered_map<std::string, int> seen;
    for (auto &nm : names)
        if (!index_.count(nm) && !seen.count(nm)) {
            seen[nm] = 1;
            fresh.push_back(&nm);
        }
    grow(((int64_t)index_.size() + (int64_t)fresh.size()) * blocks_ + scratch_rows);
    for (auto *nm : fresh) {
        const int r = (int)index_.size() * blocks_;
        index_[*nm] = r;
    }
}

int SiEncWireMap::scratch(int64_t rows)
{
    const int64_t base = (int64_t)index_.size() * blocks_;
    grow(base + rows);
    return (int)base;
}

std::vector<uint64_t> SiEncWireMap::get(const std::string &k) const
{
    const int32_t r = row(k);
    std::vector<int32_t> idx((size_t)blocks_);
    for (int b = 0; b < blocks_; b++) idx[(size_t)b] = r + b;
    std::vector<uint64_t> out((size_t)blocks_ * (dim_ + 1));
    si_ok(helm_si_wires_download(ctx_, wires_, idx.data(), out.data(), blocks_), "wires_download");
    return out;
}

void SiEncWireMap::insert(const std::string &k, const uint64_t *lwe)
{
    auto it = index_.find(k);
    int32_t r;
    if (it == index_.end()) {
        r = (int32_t)index_.size() * blocks_;
        grow(r + blocks_);
        index_[k] = r;
    } else
        r = it->second;
    std::vector<int32_t> idx((size_t)blocks_);
    for (int b = 0; b < blocks_; b++) idx[(size_t)b] = r + b;
    si_ok(helm_si_wires_upload(ctx_, wires_, idx.data(), lwe, blocks_), "wires_upload");
    gen_++;
}

std::unique_ptr<SiEncWireMap> SiEncWireMap::clone(int64_t scratch_rows) const
{
    auto m = std::make_unique<SiEncWireMap>(ctx_, blocks_);
    const int64_t used = (int64_t)index_.size() * blocks_;
    m->grow(used + scratch_rows);
    if (used > 0) {
        std::vector<int32_t> idx((size_t)used);
        for (int64_t i = 0; i < used; i++) idx[(size_t)i] = (int32_t)i;
        si_ok(helm_si_wires_copy(ctx_, wires_, idx.data(), m->wires_, idx.data(), used), "wires_copy");
    }
    m->index_ = index_;
    return m;
}

// ---------------------------------------------------------------------------------------
// LutCircuit
// ---------------------------------------------------------------------------------------
LutCircuit::LutCircuit(helm_si_client_key *client_key, helm_si_ctx *server_key, Circuit circuit)
    : client_key_(client_key), server_key_(server_key), circuit_(std::move(circuit))
{
    si_ok(helm_si_get_params(server_key, &P_), "get_params");
}

// reference src/circuit.rs:970-1000
std::unique_ptr<SiEncWireMap> LutCircuit::encrypt_inputs(const std::set<std::string> &wire_set,
                                                         const std::map<std::string, PtxtType> &input_wire_map)
{
    auto m = std::make_unique<SiEncWireMap>(server_key_, 1);
    std::vector<std::string> names(wire_set.begin(), wire_set.end());
    names.insert(names.end(), circuit_.input_wires().begin(), circuit_.input_wires().end());
    names.insert(names.end(), circuit_.dff_outputs().begin(), circuit_.dff_outputs().end());
    m->reserve_keys(names, 0);
    {
        std::vector<int32_t> idx;
        for (auto &w : wire_set) idx.push_back(m->row(w));
        std::vector<uint64_t> zeros(idx.size(), 0);
        if (!idx.empty())
            si_ok(helm_si_wires_set_trivial(server_key_, m->table(), idx.data(), zeros.data(), (int64_t)idx.size()),
                  "wires_set_trivial");
    }
    std::vector<int32_t> idx;
    std::vector<uint64_t> vals;
    const bool dummy = input_wire_map.empty() || input_wire_map.count("dummy");
    for (auto &input_wire : circuit_.input_wires()) {
        uint64_t v = 0;
        if (!dummy) {
            auto it = input_wire_map.find(input_wire);
            if (it == input_wire_map.end()) throw Panic("\n Input wire \"" + input_wire + "\" not found in input wires!");
            if (it->second.kind != PtxtType::Bool) throw Panic("internal error: entered unreachable code");
            v = it->second.as_bool() ? 1 : 0;
        }
        idx.push_back(m->row(input_wire));
        vals.push_back(v);
    }
    // DFF outputs are input wires too; the reference's second insert (encrypt(0), circuit.rs:995-997)
    // wins.  One upload must not carry the same row twice (rows are written concurrently).
    for (auto &w : circuit_.dff_outputs()) {
        const int32_t r = m->row(w);
        size_t q = 0;
        while (q < idx.size() && idx[q] != r) q++;
        if (q < idx.size()) vals[q] = 0;
        else {
            idx.push_back(r);
            vals.push_back(0);
        }
    }
    if (!idx.empty()) {
        const size_t row = (size_t)P_.k * P_.N + 1;
        std::vector<uint64_t> cts(idx.size() * row);
        if (helm_si_client_encrypt(client_key_, vals.data(), (int64_t)vals.size(), cts.data())) throw Panic(client_key_ ? "encrypt failed" : "evaluation-only circuit (no client key): encrypt with the caller's keys and insert the ciphertext words");
        si_ok(helm_si_wires_upload(server_key_, m->table(), idx.data(), cts.data(), (int64_t)idx.size()), "wires_upload");
    }
    return m;
}

// reference src/circuit.rs:1002-1011
std::unique_ptr<SiEncWireMap> LutCircuit::init_ready()
{
    auto m = std::make_unique<SiEncWireMap>(server_key_, 1);
    m->reserve_keys(circuit_.output_wires(), 0);
    std::vector<int32_t> idx;
    for (auto &w : circuit_.output_wires()) idx.push_back(m->row(w));
    std::vector<uint64_t> zeros(idx.size(), 0);
    if (!idx.empty())
        si_ok(helm_si_wires_set_trivial(server_key_, m->table(), idx.data(), zeros.data(), (int64_t)idx.size()),
              "wires_set_trivial");
    return m;
}

// reference src/circuit.rs:1013-1030: valid = enc * READY + valid * (1 - READY), here one
// 3-input look-up (enc, valid, READY) -> READY ? enc : valid per output.
void LutCircuit::evaluate_ready(const SiEncWireMap &enc_wire_map, SiEncWireMap &valid_outputs)
{
    std::vector<std::string> keys;
    for (auto &k : valid_outputs.keys())
        if (enc_wire_map.contains_key(k)) keys.push_back(k);
    if (keys.empty()) return;
    if (!enc_wire_map.contains_key("READY")) throw Panic("called `Option::unwrap()` on a `None` value (READY)");
    SiEncWireMap tmp(server_key_, 1);
    std::vector<std::string> names = {"READY"};
    for (auto &k : keys) {
        names.push_back("t:" + k);
        names.push_back("e:" + k);
    }
    tmp.reserve_keys(names, 0);
    tmp.insert("READY", enc_wire_map.get("READY").data());
    std::vector<int32_t> arity, in_idx, out;
    std::vector<uint64_t> table;
    for (auto &k : keys) {
        tmp.insert("t:" + k, enc_wire_map.get(k).data());
        tmp.insert("e:" + k, valid_outputs.get(k).data());
        arity.push_back(3);
        in_idx.push_back(tmp.row("t:" + k));
        in_idx.push_back(tmp.row("e:" + k));
        in_idx.push_back(tmp.row("READY"));
        table.push_back(0xE4); // index = enc*4 + valid*2 + READY
        out.push_back(tmp.row("e:" + k));
    }
    si_ok(helm_si_eval_lut_level(server_key_, tmp.table(), arity.data(), in_idx.data(), 3, table.data(), out.data(),
                                 (int64_t)arity.size()),
          "eval_lut_level");
    for (auto &k : keys) valid_outputs.insert(k, tmp.get("e:" + k).data());
}

void LutCircuit::set_wide_lut_key(helm_wop_ctx *wop, int bits_per_block)
{
    if (wop) {
        helm_wop_params W{};
        si_ok(helm_wop_get_params(wop, &W), "wop_get_params");
        // generate_high_precision_lut_radix_helm indexes the gate's table with sum block_j * message_modulus^j
        // (gates.rs:845-848): for inputs that hold one bit each that is the LUT index only when the basis is 2 -
        // the encoding the reference's LUT mode names (helm.rs:301); a larger basis runs past the table (a panic there)
        if (W.message_modulus != 2)
            throw Panic("wide LUT gates need message_modulus = 2 (the table rule of gates.rs:845-848)");
        if (bits_per_block < 1) throw Panic("bits_per_block must be positive");
    }
    wop_ = wop;
    wop_bits_per_block_ = bits_per_block;
    memo_.valid = false; // wide gates now take another path
}

// reference src/circuit.rs:1032-1083
std::unique_ptr<SiEncWireMap> LutCircuit::evaluate_encrypted(const SiEncWireMap &enc_wire_map, size_t cycle,
                                                             const std::string & /*ptxt_type*/)
{
    if (!circuit_.gates_empty()) throw Panic("assertion failed: self.circuit.gates.is_empty()");
    if (!circuit_.get_ordered_gates().empty()) throw Panic("assertion failed: self.circuit.ordered_gates.is_empty()");
    // same-cycle memo (gates.rs:288-292): this cycle was already evaluated on this very map -> no launch
    if (memo_.hit(cycle, enc_wire_map)) {
        memo_hits_++;
        log_ += "  Cycle " + std::to_string(cycle) + " already evaluated on these inputs: cached wire map returned\n";
        return memo_.out->clone(0);
    }
    auto eval_values = enc_wire_map.clone(0);
    const size_t total_levels = circuit_.level_map().size();
    pbs_count_ = 0;
    int capacity = 0; // index bits one block holds: gates::lut() packs sum in_i << (arity-1-i) into a single block
    while ((2 << capacity) <= P_.message_modulus * P_.carry_modulus) capacity++;
    // A level is one batched call.  Where a gate reads a wire ANOTHER gate of the level writes (flip-flops fed by
    // flip-flops: every DFF sits in the last level, circuit.rs:174-239) the reference's par_iter is a race; this
    // repository's plaintext evaluator gives a level snapshot semantics (every gate reads the values from before the level,
    // as flip-flops on one clock edge do) and the encrypted evaluation follows it: such rows are copied to scratch rows by a
    // call of copy gates in front and the readers read the copies (helm_si_eval_lut_level refuses a read-after-write inside
    // a level).  `redirect`: wire row -> scratch row for the gates of the level at hand.
    std::vector<std::pair<size_t, std::vector<Gate>>> parts; // (circuit level, gates of one call)
    for (auto &kv : circuit_.level_map()) parts.push_back({kv.first, kv.second});
    std::vector<char> last_of_level(parts.size(), 1);
    std::map<int32_t, int32_t> redirect;
    for (size_t part = 0; part < parts.size(); part++) {
        const std::pair<size_t, std::vector<Gate>> &kv = parts[part];
        const auto &gates = kv.second;
        redirect.clear();
        {
            std::unordered_map<int32_t, const Gate *> writer;
            for (auto &g : gates) writer.emplace(eval_values->row(g.get_output_wire()), &g);
            std::vector<int32_t> hazard;
            for (auto &g : gates)
                for (auto &w : g.get_input_wires()) {
                    auto it = writer.find(eval_values->row(w));
                    if (it != writer.end() && it->second != &g) hazard.push_back(it->first);
                }
            std::sort(hazard.begin(), hazard.end());
            hazard.erase(std::unique(hazard.begin(), hazard.end()), hazard.end());
            if (!hazard.empty()) {
                const int32_t base = eval_values->scratch((int64_t)hazard.size());
                std::vector<int32_t> car(hazard.size(), 0), cin(hazard), cout(hazard.size());
                std::vector<uint64_t> ctab(hazard.size(), 0);
                for (size_t q = 0; q < hazard.size(); q++) redirect[hazard[q]] = cout[q] = base + (int32_t)q;
                si_ok(helm_si_eval_lut_level(server_key_, eval_values->table(), car.data(), cin.data(), 1, ctab.data(), cout.data(),
                                             (int64_t)hazard.size()),
                      "eval_lut_level");
            }
        }
        auto in_row_of = [&](const std::string &w) {
            const int32_t r = eval_values->row(w);
            auto it = redirect.find(r);
            return it == redirect.end() ? r : it->second;
        };
        int max_in = 1;
        for (auto &g : gates)
            if (!(wop_ && g.get_gate_type() == GateType::Lut && (int)g.get_input_wires().size() > capacity))
                max_in = std::max<int>(max_in, (int)g.get_input_wires().size());
        std::vector<int32_t> arity, in_idx, out;
        std::vector<uint64_t> table;
        // wide gates of the level, grouped by input count: in_idx rows, tables, out rows
        struct Wide {
            std::vector<int32_t> in, out;
            std::vector<uint64_t> tables;
        };
        std::map<int, Wide> wide;
        for (size_t gi = 0; gi < gates.size(); gi++) {
            const Gate &g = gates[gi];
            const auto &ins = g.get_input_wires();
            if (g.get_gate_type() == GateType::Lut) {
                if (!g.get_lut_const()) throw Panic("Lut const not provided");
                if (wop_ && (int)ins.size() > capacity) {
                    helm_wop_params W{};
                    si_ok(helm_wop_get_params(wop_, &W), "wop_get_params");
                    const int m = (int)ins.size();
                    Wide &wg = wide[m];
                    for (auto &w : ins) wg.in.push_back(in_row_of(w));
                    wg.out.push_back(eval_values->row(g.get_output_wire()));
                    const size_t words = helm_wop_table_words(&W, m * wop_bits_per_block_);
                    wg.tables.resize(wg.tables.size() + words);
                    si_ok(helm_wop_make_table(&W, m, wop_bits_per_block_, g.get_lut_const()->data(),
                                              g.get_lut_const()->size(), wg.tables.data() + wg.tables.size() - words),
                          "wop_make_table");
                    // cleaning + bit removal + circuit bootstraps + the final one
                    pbs_count_ += m + m * (wop_bits_per_block_ - 1) + (int64_t)m * wop_bits_per_block_ * W.cbs_l + 1;
                    continue;
                }
                if (ins.size() > 6) throw Panic("LUT with more than 6 inputs does not fit the truth-table word");
                uint64_t bits = 0;
                for (size_t i = 0; i < g.get_lut_const()->size() && i < 64; i++)
                    if ((*g.get_lut_const())[i] & 1) bits |= 1ull << i;
                arity.push_back((int32_t)ins.size());
                table.push_back(bits);
                if (ins.size() >= 2) pbs_count_++;
            } else { // evaluate_encrypted_dff: the output is a copy of the first input (circuit.rs:1067)
                if (ins.empty()) throw Panic("gate \"" + g.get_gate_name() + "\" has no input");
                arity.push_back(0);
                table.push_back(0);
            }
            const size_t slot = in_idx.size();
            in_idx.resize(slot + (size_t)max_in, -1);
            for (size_t q = 0; q < ins.size(); q++) in_idx[slot + q] = in_row_of(ins[q]);
            out.push_back(eval_values->row(g.get_output_wire()));
        }
        const auto level_start = std::chrono::steady_clock::now();
        // wide gates first: they read the level's inputs before a state copy of the same level overwrites one
        for (auto &wk : wide)
            si_ok(helm_wop_eval_luts(wop_, eval_values->table(), wk.second.in.data(), wk.first, wop_bits_per_block_,
                                     wk.second.tables.data(), wk.second.out.data(), (int64_t)wk.second.out.size()),
                  "wop_eval_luts");
        if (!arity.empty())
            si_ok(helm_si_eval_lut_level(server_key_, eval_values->table(), arity.data(), in_idx.data(), max_in,
                                         table.data(), out.data(), (int64_t)arity.size()),
                  "eval_lut_level");
        std::ostringstream os;
        // gates.rs:293-302 prints the time of every gate's lut() call; the gates of a level are one batched dispatch
        // here, so each of them took the level's time.  That needs one host synchronisation per level, which also keeps
        // the host from preparing the next level meanwhile: set_timing_lines(false) drops both.
        if (timing_lines_) {
            si_ok(helm_si_sync(server_key_), "sync");
            const auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - level_start).count();
            for (auto &g : gates)
                if (g.get_gate_type() == GateType::Lut) os << "PBS time: " << us << " us\n";
        }
        if (last_of_level[part]) os << "  Evaluated gates in level [" << kv.first << "/" << total_levels << "]\n";
        append_log(log_, os.str());
    }
    si_ok(helm_si_sync(server_key_), "sync");
    memo_.out = eval_values->clone(0);
    memo_.cycle = cycle;
    memo_.in_id = enc_wire_map.id();
    memo_.in_gen = enc_wire_map.generation();
    memo_.valid = true;
    return eval_values;
}

// reference src/circuit.rs:1085-1110
std::map<std::string, PtxtType> LutCircuit::decrypt_outputs(const SiEncWireMap &enc_wire_map, bool verbose)
{
    std::map<std::string, PtxtType> out;
    for (auto &w : circuit_.output_wires()) {
        auto ct = enc_wire_map.get(w);
        uint64_t v = 0;
        if (helm_si_client_decrypt(client_key_, ct.data(), 1, &v)) throw Panic(client_key_ ? "decrypt failed" : "evaluation-only circuit (no client key): read the ciphertext words and decrypt with the caller's keys");
        PtxtType p;
        p.kind = PtxtType::U64;
        p.value = v % (uint64_t)P_.message_modulus; // ClientKey::decrypt: the message part
        out[w] = p;
    }
    size_t i = 0;
    for (auto &kv : out) {
        if (i > 10 && !verbose) {
            log_ += "[!] More than ten output_wires, pass `--verbose` to see output.\n";
            break;
        }
        log_ += " " + kv.first + ": " + kv.second.to_string() + "\n";
        i++;
    }
    return out;
}

// ---------------------------------------------------------------------------------------
// RadixEngine: level-batched radix-integer operators on rows of one ciphertext table.
// An integer = `nb` consecutive rows (block 0 least significant), every block a shortint
// with a 2-bit message (message_modulus 4) and value below 16.
// ---------------------------------------------------------------------------------------
RadixEngine::RadixEngine(helm_si_ctx *ctx, int nb) : ctx_(ctx), nb_(nb)
{
    si_ok(helm_si_get_params(ctx, &P_), "get_params");
    if (P_.message_modulus != 4 || P_.carry_modulus != 4)
        throw Panic("the radix layer needs message_modulus = carry_modulus = 4 (PARAM_MESSAGE_2_CARRY_2)");
    const int t = 16;
    auto add_lut = [&](auto f) {
        std::vector<uint64_t> vals((size_t)t);
        for (int v = 0; v < t; v++) vals[(size_t)v] = (uint64_t)f(v);
        const size_t at = luts_.size();
        luts_.resize(at + (size_t)P_.N);
        si_ok(helm_si_make_lut(ctx_, vals.data(), luts_.data() + at), "make_lut");
        return (int)(at / (size_t)P_.N);
    };
    lut_msg_ = add_lut([](int v) { return v & 3; });
    lut_carry_ = add_lut([](int v) { return v >> 2; });
    // carry propagation (see propagate()): values are taken mod 32 - 16 is the padding bit, 32 - x is -x
    for (int p = 0; p < 4; p++) { // carry state of a block sum x <= 7 (0 absorbs, 1 propagates, 2 generates), weighted 2^p
        lut_t_[p] = add_lut([p](int v) { return (v >= 4 ? 2 : v == 3 ? 1 : 0) << p; });
        // state of a whole group from V_3 = t_0 + 2 t_1 + 4 t_2 + 8 t_3 <= 30, negacyclic: -2^p (+ 2^p) below 15, 0 at 15
        lut_s_[p] = add_lut([p](int v) { return v == 15 ? 0 : 32 - (1 << p); });
    }
    for (int i = 0; i < 3; i++) { // prefix 0..i of a group from V_i: state x 4, and the bare carry (x 4)
        lut_q_[i] = add_lut([i](int v) { return v >= (2 << i) ? 8 : v == (2 << i) - 1 ? 4 : 0; });
        lut_gc_[i] = add_lut([i](int v) { return v >= (2 << i) ? 4 : 0; });
    }
    lut_q3_ = add_lut([](int v) { return v == 15 ? 0 : 32 - 4; }); // + 4 afterwards: 0 / 4 / 8
    lut_gc3_ = add_lut([](int) { return 32 - 2; });                // + 2 afterwards: 0 below 16, 4 from 16 on
    lut_resolve_ = add_lut([](int v) { return v >= 8 ? 4 : 0; });  // c form (4 c) -> bit form
    lut_final_ = add_lut([](int v) { return ((v & 3) + ((v >> 2) >= 2 ? 1 : 0)) & 3; });
    lut_cout_ = add_lut([](int v) { return v >= 8 ? 1 : 0; });
    // products of two 2-bit messages, packed a * 4 + b
    lut_mul_lo_ = add_lut([](int v) { return ((v >> 2) * (v & 3)) & 3; });
    lut_mul_hi_ = add_lut([](int v) { return ((v >> 2) * (v & 3)) >> 2; });
    // both cross products of a square at once: 2 * a_j * a_k
    lut_mul2_lo_ = add_lut([](int v) { return (2 * (v >> 2) * (v & 3)) & 3; });
    lut_mul2_hi_ = add_lut([](int v) { return (2 * (v >> 2) * (v & 3)) >> 2; });
    lut_bit0_ = add_lut([](int v) { return v & 1; });
    lut_bit1_ = add_lut([](int v) { return (v >> 1) & 1; });
    // one-bit shifts of a block x with its neighbour y, packed 4 * x + y
    lut_shl1_ = add_lut([](int v) { return (((v >> 2) << 1) & 3) | ((v & 3) >> 1); });
    lut_shr1_ = add_lut([](int v) { return ((v >> 2) >> 1) | (((v & 3) & 1) << 1); });
    lut_sel_ = add_lut([](int v) { return (v >> 2) ? (v & 3) : 0; }); // 4 * c + x -> c ? x : 0
}

// ---------------------------------------------------------------------------------------
// RoundMerger
// ---------------------------------------------------------------------------------------
RoundMerger::RoundMerger(helm_si_ctx *ctx, int chains, int64_t capacity, bool strict)
    : ctx_(ctx), capacity_(std::max<int64_t>(capacity, 1)), strict_(strict), subs_((size_t)chains), remaining_((size_t)chains, 0),
      active_((size_t)chains, 1)
{
}

// mu_ held.  Launches while every running chain waits with a round; returns as soon as one of them has been released.
void RoundMerger::issue_locked()
{
    for (;;) {
        size_t running = 0, present = 0;
        int64_t total = 0;
        for (size_t c = 0; c < subs_.size(); c++) {
            running += active_[c] ? 1 : 0;
            if (subs_[c].present) {
                present++;
                total += (int64_t)(subs_[c].in->size() - subs_[c].taken);
            }
        }
        if (present == 0 || present != running) return;
        std::vector<size_t> order;
        for (size_t c = 0; c < subs_.size(); c++)
            if (subs_[c].present) order.push_back(c);
        std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return remaining_[x] > remaining_[y]; });
        // a single chain left, or everything fits: one launch; otherwise fill the device once, most urgent chain first
        // (strict_: a capacity the caller set is honoured for a lone chain too - the tests' way to cut everywhere)
        const int64_t cap = ((present == 1 && !strict_) || total <= capacity_) ? total : capacity_;
        std::vector<int32_t> in, lut, out;
        in.reserve((size_t)cap), lut.reserve((size_t)cap), out.reserve((size_t)cap);
        const Sub &first = subs_[order[0]];
        for (size_t c : order) {
            Sub &s = subs_[c];
            if (s.w != first.w || s.n_luts != first.n_luts) {
                error_ = "round merger: chains on different wire tables or look-up tables";
                break;
            }
            const size_t n = (size_t)std::min<int64_t>((int64_t)(s.in->size() - s.taken), cap - (int64_t)in.size());
            in.insert(in.end(), s.in->begin() + (long)s.taken, s.in->begin() + (long)(s.taken + n));
            lut.insert(lut.end(), s.lut->begin() + (long)s.taken, s.lut->begin() + (long)(s.taken + n));
            out.insert(out.end(), s.out->begin() + (long)s.taken, s.out->begin() + (long)(s.taken + n));
            s.taken += n;
        }
        if (error_.empty() && !in.empty()) {
            if (helm_si_apply_luts(ctx_, first.w, in.data(), lut.data(), out.data(), (int64_t)in.size(), first.luts, first.n_luts) != 0)
                error_ = std::string("apply_luts: ") + helm_hip_last_error();
            launches_++;
        }
        bool released = false;
        for (size_t c : order) {
            Sub &s = subs_[c];
            if (!error_.empty() || s.taken == s.in->size()) {
                s.present = false;
                if (remaining_[c] > 0) remaining_[c]--;
                released = true;
            }
        }
        if (released) {
            cv_.notify_all();
            return;
        }
    }
}

void RoundMerger::submit(int chain, helm_si_wires *w, const std::vector<int32_t> &in, const std::vector<int32_t> &lut,
                         const std::vector<int32_t> &out, const uint64_t *luts, int64_t n_luts)
{
    // Ordering contract of a round (RadixEngine::apply): the merger may cut a round into several launches, so the
    // guarantee of ONE helm_si_apply_luts call - every keyswitch reads its input before any bootstrap writes - only holds
    // inside each part.  A round is therefore safe to cut anywhere iff no entry reads a row an EARLIER-listed entry writes
    // (readers of a row come before its in-place writer; an entry may rewrite its own input row).  Checked here, for every
    // round, whether or not this device's capacity happens to cut it: a batch that lists a writer first would otherwise
    // return wrong ciphertexts on some devices only.
    {
        std::unordered_set<int32_t> written;
        written.reserve(out.size() * 2);
        for (size_t i = 0; i < in.size(); i++) {
            if (written.count(in[i]))
                throw Panic("round merger: entry " + std::to_string(i) + " of a look-up round reads row " + std::to_string(in[i]) +
                            ", which an earlier entry of the same round writes - list the readers of a row before its in-place writer");
            written.insert(out[i]);
        }
    }
    std::unique_lock<std::mutex> lk(mu_);
    if (!error_.empty()) throw Panic(error_);
    Sub &s = subs_[(size_t)chain];
    s.w = w, s.in = &in, s.lut = &lut, s.out = &out, s.luts = luts, s.n_luts = n_luts, s.taken = 0, s.present = true;
    issue_locked();
    cv_.wait(lk, [&] { return !s.present; });
    if (!error_.empty()) throw Panic(error_);
}

void RoundMerger::finish(int chain)
{
    std::unique_lock<std::mutex> lk(mu_);
    active_[(size_t)chain] = 0;
    subs_[(size_t)chain].present = false;
    issue_locked(); // the others may all be waiting now
}

void RadixEngine::lincomb(helm_si_wires *w, const std::vector<int32_t> &in_idx, const std::vector<int64_t> &coef,
                          const std::vector<int64_t> &cadd, const std::vector<int32_t> &out, int terms)
{
    if (out.empty()) return;
    auto guard = device_lock();
    si_ok(helm_si_lincomb(ctx_, w, in_idx.data(), coef.data(), cadd.empty() ? nullptr : cadd.data(), out.data(), terms,
                          (int64_t)out.size()),
          "lincomb");
}

void RadixEngine::apply(helm_si_wires *w, const std::vector<int32_t> &in, const std::vector<int32_t> &lut,
                        const std::vector<int32_t> &out)
{
    // look-ups that wait for a batch to ride in (shift_scalar): independent rows, so any batch of the level will do
    if (!pend_in_.empty()) {
        std::vector<int32_t> i2(in), l2(lut), o2(out);
        i2.insert(i2.end(), pend_in_.begin(), pend_in_.end());
        l2.insert(l2.end(), pend_lut_.begin(), pend_lut_.end());
        o2.insert(o2.end(), pend_out_.begin(), pend_out_.end());
        pend_in_.clear();
        pend_lut_.clear();
        pend_out_.clear();
        apply(w, i2, l2, o2);
        return;
    }
    if (in.empty()) return;
    if (merger_)
        merger_->submit(chain_, w, in, lut, out, luts_.data(), (int64_t)(luts_.size() / (size_t)P_.N));
    else
        si_ok(helm_si_apply_luts(ctx_, w, in.data(), lut.data(), out.data(), (int64_t)in.size(), luts_.data(),
                                 (int64_t)(luts_.size() / (size_t)P_.N)),
              "apply_luts");
    pbs_count_ += (int64_t)in.size();
    pbs_rounds_++;
}

// Full carry propagation of integers of `W` blocks whose block sums are <= 6 (block 0: <= 7), in
//     2 + ceil(log4 W) rounds of look-ups (+1 per level beyond two: W > 16)
// instead of the 2 + log2 W of a Hillis-Steele prefix over the carry states - four rounds instead of six for a u32.
// A block's carry state t is 0 (absorbs), 1 (propagates an incoming carry: sum == 3) or 2 (generates: sum >= 4).  The
// states of a group of four blocks, weighted 1, 2, 4, 8 BY THEIR LOOK-UP TABLES (a bootstrap's output noise does not
// depend on the table, so the weights cost nothing), add up like the operands of a binary adder:
//     V_i = sum_{j <= i} t_j 2^j   carries out of bit i  <=>  blocks 0..i generate,   V_i == 2^(i+1) - 1  <=>  they propagate
// so ONE look-up on V_i gives the state of the prefix 0..i of the group (Q_i), for every position of the group in the same
// round.  V_3 needs five bits: its table uses the padding bit (negacyclic look-up: f(v + 16) = -f(v); f = -x below 15,
// 0 at 15, and a plaintext + x afterwards gives 0 / x / 2x for absorb / propagate / generate; V_3 <= 30).  The group
// states S_k, again weighted by their tables, go through the same step one level up (recursively for more than 16
// blocks), which yields the carry gc_k into every group; the carry into block i of group k is then c = Q_(i-1) + gc_k >= 2,
// and the last round computes (message + [c >= 2]) & 3 from the packed value message + 4 c.  Every look-up input is a sum
// of at most four fresh ciphertexts.  (tfhe's own radix propagation lives in the absent crate; results mod 2^bits are what
// the reference's tests pin: tests/gates_test.rs:127-310.)
// `scratch` needs prop_rows(W) rows per integer.  With `flags` (one row per integer) the carry OUT of the top block is
// kept there (0 / 1) instead of being dropped.
int RadixEngine::prop_rows(int W) { return 5 * W + 16; }

// Carries INTO items 1..need of `n` items per integer (G integers) whose states sit in st[g * n + m], weighted 2^(m % 4);
// add_const: the states still lack the plaintext + 2^(m % 4) of the negacyclic table (group states).  Returns rows
// [g * (need + 1) + m]; bit form (0 / 4) if `bit`, else c form (4 c, carry <=> c >= 2).
RadixEngine::Carries RadixEngine::carries(helm_si_wires *w, const std::vector<int32_t> &st, int G, int n, int need,
                                         bool add_const, int &sp, bool want_bit)
{
    auto take = [&](int rows) { const int b = sp; sp += rows; return b; };
    Carries R;
    R.bit = true;
    R.row.assign((size_t)G * (size_t)(need + 1), -1);
    if (need <= 0) return R;
    std::vector<int32_t> li, lo, in, lut, out, fix_rows;
    std::vector<int64_t> lc, ca, fix_c;
    auto sum_into = [&](int g, int first, int count, int dst) { // dst = sum of items first..first+count-1 (count <= 4)
        int64_t c = 0;
        for (int j = 0; j < 4; j++) {
            li.push_back(j < count ? st[(size_t)g * n + first + j] : -1);
            lc.push_back(j < count ? 1 : 0);
            if (j < count && add_const) c += (int64_t)1 << ((first + j) % 4);
        }
        lo.push_back(dst);
        ca.push_back(c);
    };
    if (n <= 4) {
        const int base = take(G * need);
        for (int g = 0; g < G; g++)
            for (int m = 1; m <= need; m++) {
                const int r = base + g * need + (m - 1);
                sum_into(g, 0, m, r);
                in.push_back(r), out.push_back(r), lut.push_back(m - 1 < 3 ? lut_gc_[m - 1] : lut_gc3_);
                if (m - 1 == 3) fix_rows.push_back(r), fix_c.push_back(2);
                R.row[(size_t)g * (need + 1) + m] = r;
            }
        lincomb(w, li, lc, ca, lo, 4);
        apply(w, in, lut, out);
        R.bit = true;
    } else {
        const int ng = (n + 3) / 4;
        // one round: prefix states inside every group, group states of the full groups that are not the last
        const int vbase = take(G * n), sbase = take(G * ng);
        std::vector<int32_t> s_items((size_t)G * ng, -1), in_q, lut_q, out_q;
        for (int g = 0; g < G; g++)
            for (int k = 0; k < ng; k++) {
                const int cnt = std::min(4, n - 4 * k);
                for (int i = 0; i < cnt; i++) {
                    const int m = 4 * k + i + 1; // the item this prefix carries into
                    const bool want_q = m <= need && (m % 4 != 0 || m == n);
                    const bool want_s = i == 3 && k < ng - 1;
                    if (!want_q && !want_s) continue;
                    const int v = vbase + g * n + 4 * k + i;
                    sum_into(g, 4 * k, i + 1, v);
                    if (want_s) { // reads v before the in-place look-up below may overwrite it: listed first
                        const int srow = sbase + g * ng + k;
                        in.push_back(v), out.push_back(srow), lut.push_back(lut_s_[k % 4]);
                        s_items[(size_t)g * ng + k] = srow;
                    }
                    if (want_q) {
                        in_q.push_back(v), out_q.push_back(v), lut_q.push_back(i < 3 ? lut_q_[i] : lut_q3_);
                        if (i == 3) fix_rows.push_back(v), fix_c.push_back(4);
                    }
                }
            }
        lincomb(w, li, lc, ca, lo, 4);
        in.insert(in.end(), in_q.begin(), in_q.end());
        lut.insert(lut.end(), lut_q.begin(), lut_q.end());
        out.insert(out.end(), out_q.begin(), out_q.end());
        apply(w, in, lut, out);
        if (!fix_rows.empty()) { // the plaintext halves of the negacyclic tables
            std::vector<int64_t> fc(fix_rows.size(), 1);
            lincomb(w, fix_rows, fc, fix_c, fix_rows, 1);
            fix_rows.clear(), fix_c.clear();
        }
        // carries into the groups, one level up (bit form: they are added to the prefix states)
        const int gneed = (need == n && n % 4 == 0) ? n / 4 - 1 : need / 4; // the last group a needed item sits in
        Carries GC = carries(w, s_items, G, ng, gneed, true, sp, true);
        // c form of every item: prefix state of the blocks below it in its group + the carry into the group
        li.clear(), lc.clear(), lo.clear(), ca.clear();
        const int cbase = take(G * need);
        for (int g = 0; g < G; g++)
            for (int m = 1; m <= need; m++) {
                const int k = (m == n && m % 4 == 0) ? m / 4 - 1 : m / 4; // the virtual item behind a full last group
                const int gc = k >= 1 ? GC.row[(size_t)g * (gneed + 1) + k] : -1;
                const bool pos0 = m % 4 == 0 && m != n;
                const int q = pos0 ? -1 : vbase + g * n + (m - 1);
                const int r = cbase + g * need + (m - 1);
                li.push_back(q), lc.push_back(q >= 0 ? 1 : 0);
                li.push_back(gc), lc.push_back(gc >= 0 ? 1 : 0);
                lo.push_back(r);
                ca.push_back(pos0 ? 4 : 0); // first block of a group: c = 1 + gc
                R.row[(size_t)g * (need + 1) + m] = r;
            }
        lincomb(w, li, lc, ca, lo, 2);
        R.bit = false;
        if (want_bit) {
            in.clear(), lut.clear(), out.clear();
            for (int g = 0; g < G; g++)
                for (int m = 1; m <= need; m++) {
                    const int r = R.row[(size_t)g * (need + 1) + m];
                    in.push_back(r), out.push_back(r), lut.push_back(lut_resolve_);
                }
            apply(w, in, lut, out);
            R.bit = true;
        }
    }
    if (!fix_rows.empty()) {
        std::vector<int64_t> fc(fix_rows.size(), 1);
        lincomb(w, fix_rows, fc, fix_c, fix_rows, 1);
    }
    return R;
}

void RadixEngine::propagate(helm_si_wires *w, const std::vector<int32_t> &bases, int scratch, int W,
                            const std::vector<int32_t> *flags)
{
    const int G = (int)bases.size();
    if (G == 0) return;
    if (W <= 0) W = nb_;
    int sp = scratch;
    auto take = [&](int rows) { const int b = sp; sp += rows; return b; };
    const int need = flags ? W : W - 1; // carries into blocks 1..W-1, and out of the top block
    // round 1: weighted carry states, and the messages (in place)
    const int tbase = take(G * W);
    std::vector<int32_t> st((size_t)G * W), in, lut, out, in2, lut2, out2;
    for (int g = 0; g < G; g++)
        for (int i = 0; i < W; i++) {
            st[(size_t)g * W + i] = tbase + g * W + i;
            if (i < need) in.push_back(bases[(size_t)g] + i), lut.push_back(lut_t_[i % 4]), out.push_back(tbase + g * W + i);
            in2.push_back(bases[(size_t)g] + i), lut2.push_back(lut_msg_), out2.push_back(bases[(size_t)g] + i);
        }
    in.insert(in.end(), in2.begin(), in2.end());
    lut.insert(lut.end(), lut2.begin(), lut2.end());
    out.insert(out.end(), out2.begin(), out2.end());
    apply(w, in, lut, out);
    if (need == 0) return;
    Carries C = carries(w, st, G, W, need, false, sp, false);
    // last round: message + 4 c -> (message + [c >= 2]) & 3; bit-form carries are c = 1 + bit
    std::vector<int32_t> li, lo;
    std::vector<int64_t> lc, ca;
    in.clear(), lut.clear(), out.clear();
    for (int g = 0; g < G; g++) {
        for (int i = 1; i < W; i++) {
            const int b = bases[(size_t)g] + i;
            li.push_back(b), lc.push_back(1);
            li.push_back(C.row[(size_t)g * (need + 1) + i]), lc.push_back(1);
            lo.push_back(b), ca.push_back(C.bit ? 4 : 0);
            in.push_back(b), lut.push_back(lut_final_), out.push_back(b);
        }
        if (flags) {
            const int c = C.row[(size_t)g * (need + 1) + W];
            li.push_back(c), lc.push_back(1);
            li.push_back(-1), lc.push_back(0);
            lo.push_back(c), ca.push_back(C.bit ? 4 : 0);
            in.push_back(c), lut.push_back(lut_cout_), out.push_back((*flags)[(size_t)g]);
        }
    }
    lincomb(w, li, lc, ca, lo, 2);
    apply(w, in, lut, out);
}

// Shift by a plaintext amount (src/gates.rs:602-700, scalar forms): whole blocks move for free,
// an odd amount costs one bivariate look-up per block on the pair (4 * x + y).
void RadixEngine::shift_scalar(helm_si_wires *w, const std::vector<RadixOp> &ops)
{
    std::vector<int32_t> li, lo, in, lut, out;
    std::vector<int64_t> lc;
    for (auto &op : ops) {
        const bool left = op.kind == RadixOp::ShlScalar;
        if (!left && op.kind != RadixOp::ShrScalar) continue;
        const int s = (int)(op.scalar % (unsigned)(2 * nb_)), q = s >> 1, r = s & 1;
        if (op.a2 >= 0) { // carry-save operand: whole blocks only, both terms move alike and stay in that form
            if (!left || r || op.out2 < 0) throw Panic("internal error: carry-save operand of a shift that is not by whole blocks");
            for (int i = 0; i < nb_; i++) {
                li.push_back(i >= q ? op.a2 + (i - q) : -1);
                lc.push_back(i >= q ? 1 : 0);
                li.push_back(-1);
                lc.push_back(0);
                lo.push_back(op.out2 + i);
            }
        }
        for (int i = 0; i < nb_; i++) {
            const int xi = left ? i - q : i + q, yi = left ? xi - 1 : xi + 1; // y: the neighbour bits come from
            const bool xok = xi >= 0 && xi < nb_, yok = yi >= 0 && yi < nb_;
            li.push_back(xok ? op.a + xi : -1);
            lc.push_back(xok ? (r ? 4 : 1) : 0);
            li.push_back(r && yok ? op.a + yi : -1);
            lc.push_back(r && yok ? 1 : 0);
            lo.push_back(op.out + i);
            if (r) {
                in.push_back(op.out + i);
                lut.push_back(left ? lut_shl1_ : lut_shr1_);
                out.push_back(op.out + i);
            }
        }
    }
    // operands may alias outputs (out == a): lincomb stages every sum before writing any row
    lincomb(w, li, lc, {}, lo, 2);
    // the look-ups join the level's next batch (run_level flushes what is left at its end)
    pend_in_.insert(pend_in_.end(), in.begin(), in.end());
    pend_lut_.insert(pend_lut_.end(), lut.begin(), lut.end());
    pend_out_.insert(pend_out_.end(), out.begin(), out.end());
}

// Shift by an encrypted amount (mod bits): the amount's bits select, stage by stage, between the
// value and the value shifted by 2^t (a barrel shifter); a selection is two look-ups per block,
// sel(c, x) = c ? x : 0 on 4 * c + x, summed.
void RadixEngine::shift_encrypted(helm_si_wires *w, const std::vector<RadixOp> &ops, int &sp)
{
    std::vector<const RadixOp *> sh;
    for (auto &op : ops)
        if (op.kind == RadixOp::Shl || op.kind == RadixOp::Shr) sh.push_back(&op);
    if (sh.empty()) return;
    const int G = (int)sh.size();
    int nbits = 0;
    while ((1 << nbits) < 2 * nb_) nbits++;
    auto take = [&](int rows) { const int b = sp; sp += rows; return b; };
    const int bits0 = take(G * nbits), cur0 = take(G * nb_), shf0 = take(G * nb_), selA0 = take(G * nb_), selB0 = take(G * nb_);
    auto BIT = [&](int g, int t) { return bits0 + g * nbits + t; };
    std::vector<int32_t> li, lo, in, lut, out;
    std::vector<int64_t> lc, ca;
    // bits of the amount; working copy of the value
    for (int g = 0; g < G; g++) {
        for (int t = 0; t < nbits; t++) {
            in.push_back(sh[(size_t)g]->b + (t >> 1));
            lut.push_back((t & 1) ? lut_bit1_ : lut_bit0_);
            out.push_back(BIT(g, t));
        }
        for (int i = 0; i < nb_; i++) {
            li.push_back(sh[(size_t)g]->a + i); lc.push_back(1); li.push_back(-1); lc.push_back(0);
            lo.push_back(cur0 + g * nb_ + i);
        }
    }
    lincomb(w, li, lc, {}, lo, 2);
    apply(w, in, lut, out);
    for (int t = 0; t < nbits; t++) {
        // candidate: the value shifted by 2^t bits
        li.clear(); lc.clear(); lo.clear(); in.clear(); lut.clear(); out.clear();
        for (int g = 0; g < G; g++) {
            const bool left = sh[(size_t)g]->kind == RadixOp::Shl;
            const int q = t == 0 ? 0 : 1 << (t - 1), r = t == 0 ? 1 : 0;
            for (int i = 0; i < nb_; i++) {
                const int xi = left ? i - q : i + q, yi = left ? xi - 1 : xi + 1;
                const bool xok = xi >= 0 && xi < nb_, yok = yi >= 0 && yi < nb_;
                li.push_back(xok ? cur0 + g * nb_ + xi : -1);
                lc.push_back(xok ? (r ? 4 : 1) : 0);
                li.push_back(r && yok ? cur0 + g * nb_ + yi : -1);
                lc.push_back(r && yok ? 1 : 0);
                lo.push_back(shf0 + g * nb_ + i);
                if (r) { in.push_back(shf0 + g * nb_ + i); lut.push_back(left ? lut_shl1_ : lut_shr1_); out.push_back(shf0 + g * nb_ + i); }
            }
        }
        lincomb(w, li, lc, {}, lo, 2);
        apply(w, in, lut, out);
        // cur = bit ? shifted : cur
        li.clear(); lc.clear(); lo.clear(); ca.clear(); in.clear(); lut.clear(); out.clear();
        for (int g = 0; g < G; g++)
            for (int i = 0; i < nb_; i++) {
                li.push_back(BIT(g, t)); lc.push_back(4); li.push_back(shf0 + g * nb_ + i); lc.push_back(1);
                lo.push_back(selA0 + g * nb_ + i); ca.push_back(0);
                li.push_back(BIT(g, t)); lc.push_back(-4); li.push_back(cur0 + g * nb_ + i); lc.push_back(1);
                lo.push_back(selB0 + g * nb_ + i); ca.push_back(4);
                in.push_back(selA0 + g * nb_ + i); lut.push_back(lut_sel_); out.push_back(selA0 + g * nb_ + i);
                in.push_back(selB0 + g * nb_ + i); lut.push_back(lut_sel_); out.push_back(selB0 + g * nb_ + i);
            }
        lincomb(w, li, lc, ca, lo, 2);
        apply(w, in, lut, out);
        li.clear(); lc.clear(); lo.clear();
        for (int g = 0; g < G; g++)
            for (int i = 0; i < nb_; i++) {
                li.push_back(selA0 + g * nb_ + i); lc.push_back(1); li.push_back(selB0 + g * nb_ + i); lc.push_back(1);
                lo.push_back(cur0 + g * nb_ + i);
            }
        lincomb(w, li, lc, {}, lo, 2);
    }
    li.clear(); lc.clear(); lo.clear();
    for (int g = 0; g < G; g++)
        for (int i = 0; i < nb_; i++) {
            li.push_back(cur0 + g * nb_ + i); lc.push_back(1); li.push_back(-1); lc.push_back(0);
            lo.push_back(sh[(size_t)g]->out + i);
        }
    lincomb(w, li, lc, {}, lo, 2);
}

// Unsigned division (quotient), restoring: for every numerator bit, most significant first,
// R = 2R + bit; D = R - B with its carry out (1 <=> R >= B); R = carry ? D : R; the carry is the
// quotient bit.  R has one block more than the operands (2R + 1 < 2B).  A zero divisor yields
// the all-ones quotient, as tfhe's.
void RadixEngine::divide(helm_si_wires *w, const std::vector<RadixOp> &ops, int &sp)
{
    std::vector<const RadixOp *> dv;
    for (auto &op : ops)
        if (op.kind == RadixOp::Div || op.kind == RadixOp::DivScalar) dv.push_back(&op);
    if (dv.empty()) return;
    const int G = (int)dv.size(), W = nb_ + 1, bits = 2 * nb_;
    auto take = [&](int rows) { const int b = sp; sp += rows; return b; };
    const int abit0 = take(G * bits), qbit0 = take(G * bits), B0 = take(G * W), R0 = take(G * W), D0 = take(G * W),
              selA0 = take(G * W), selB0 = take(G * W), st0 = take(prop_rows(W) * G);
    std::vector<int32_t> li, lo, in, lut, out, idx;
    std::vector<int64_t> lc, ca;
    std::vector<uint64_t> val;
    // divisor widened by one zero block (plaintext divisors as trivial ciphertexts), R = 0
    for (int g = 0; g < G; g++)
        for (int i = 0; i < W; i++) {
            const bool scalar = dv[(size_t)g]->kind == RadixOp::DivScalar;
            if (scalar || i == nb_) {
                idx.push_back(B0 + g * W + i);
                val.push_back(i < nb_ ? (uint64_t)((dv[(size_t)g]->scalar >> (2 * i)) & 3) : 0);
            } else {
                li.push_back(dv[(size_t)g]->b + i); lc.push_back(1); li.push_back(-1); lc.push_back(0);
                lo.push_back(B0 + g * W + i);
            }
            idx.push_back(R0 + g * W + i);
            val.push_back(0);
        }
    lincomb(w, li, lc, {}, lo, 2);
    {
        auto guard = device_lock();
        si_ok(helm_si_wires_set_trivial(ctx_, w, idx.data(), val.data(), (int64_t)idx.size()), "set_trivial");
    }
    // bits of the numerators
    for (int g = 0; g < G; g++)
        for (int t = 0; t < bits; t++) {
            in.push_back(dv[(size_t)g]->a + (t >> 1));
            lut.push_back((t & 1) ? lut_bit1_ : lut_bit0_);
            out.push_back(abit0 + g * bits + t);
        }
    apply(w, in, lut, out);
    std::vector<int32_t> dbase((size_t)G), flags((size_t)G);
    for (int t = bits - 1; t >= 0; t--) {
        // R = 2R + a_t : pairs (4 * R_i + R_{i-1}) -> ((x << 1) & 3) | (y >> 1), then + bit on block 0
        li.clear(); lc.clear(); lo.clear(); in.clear(); lut.clear(); out.clear();
        for (int g = 0; g < G; g++)
            for (int i = 0; i < W; i++) {
                li.push_back(R0 + g * W + i); lc.push_back(4);
                li.push_back(i > 0 ? R0 + g * W + i - 1 : -1); lc.push_back(i > 0 ? 1 : 0);
                lo.push_back(D0 + g * W + i);
                in.push_back(D0 + g * W + i); lut.push_back(lut_shl1_); out.push_back(R0 + g * W + i);
            }
        lincomb(w, li, lc, {}, lo, 2);
        apply(w, in, lut, out);
        // D = R + a_t (block 0) + ~B + 1 : block sums, then propagation with the carry out
        li.clear(); lc.clear(); lo.clear(); ca.clear();
        for (int g = 0; g < G; g++) {
            // fold the incoming bit into R first (it is needed again if the subtraction is undone)
            li.push_back(R0 + g * W); lc.push_back(1); li.push_back(abit0 + g * bits + t); lc.push_back(1);
            lo.push_back(R0 + g * W); ca.push_back(0);
        }
        lincomb(w, li, lc, ca, lo, 2);
        li.clear(); lc.clear(); lo.clear(); ca.clear();
        for (int g = 0; g < G; g++) {
            for (int i = 0; i < W; i++) {
                li.push_back(R0 + g * W + i); lc.push_back(1); li.push_back(B0 + g * W + i); lc.push_back(-1);
                lo.push_back(D0 + g * W + i); ca.push_back(i == 0 ? 4 : 3);
            }
            dbase[(size_t)g] = D0 + g * W;
            flags[(size_t)g] = qbit0 + g * bits + t;
        }
        lincomb(w, li, lc, ca, lo, 2);
        propagate(w, dbase, st0, W, &flags);
        // R = q ? D : R
        li.clear(); lc.clear(); lo.clear(); ca.clear(); in.clear(); lut.clear(); out.clear();
        for (int g = 0; g < G; g++)
            for (int i = 0; i < W; i++) {
                const int q = qbit0 + g * bits + t;
                li.push_back(q); lc.push_back(4); li.push_back(D0 + g * W + i); lc.push_back(1);
                lo.push_back(selA0 + g * W + i); ca.push_back(0);
                li.push_back(q); lc.push_back(-4); li.push_back(R0 + g * W + i); lc.push_back(1);
                lo.push_back(selB0 + g * W + i); ca.push_back(4);
                in.push_back(selA0 + g * W + i); lut.push_back(lut_sel_); out.push_back(selA0 + g * W + i);
                in.push_back(selB0 + g * W + i); lut.push_back(lut_sel_); out.push_back(selB0 + g * W + i);
            }
        lincomb(w, li, lc, ca, lo, 2);
        apply(w, in, lut, out);
        li.clear(); lc.clear(); lo.clear();
        for (int g = 0; g < G; g++)
            for (int i = 0; i < W; i++) {
                li.push_back(selA0 + g * W + i); lc.push_back(1); li.push_back(selB0 + g * W + i); lc.push_back(1);
                lo.push_back(R0 + g * W + i);
            }
        lincomb(w, li, lc, {}, lo, 2);
    }
    // quotient blocks from the quotient bits
    li.clear(); lc.clear(); lo.clear();
    for (int g = 0; g < G; g++)
        for (int i = 0; i < nb_; i++) {
            li.push_back(qbit0 + g * bits + 2 * i); lc.push_back(1);
            li.push_back(qbit0 + g * bits + 2 * i + 1); lc.push_back(2);
            lo.push_back(dv[(size_t)g]->out + i);
        }
    lincomb(w, li, lc, {}, lo, 2);
}

int64_t RadixEngine::scratch_rows(const std::vector<RadixOp> &ops) const
{
    int64_t rows = 0;
    for (auto &op : ops) {
        rows += prop_rows(nb_); // propagate scratch
        // partial-product vectors (2 nb - 1) plus the message / carry vectors of the reduction
        // rounds (about 1.5 nb): 4 nb + 4 vectors of nb rows bound both
        if (op.kind == RadixOp::Mul || op.kind == RadixOp::MulScalar) rows += (int64_t)(4 * nb_ + 4) * nb_;
        if (op.kind == RadixOp::AddScalar || op.kind == RadixOp::SubScalar) rows += nb_;
        if (op.a2 >= 0 || op.b2 >= 0 || ((op.kind == RadixOp::Add || op.kind == RadixOp::Sub) && op.out2 >= 0))
            rows += 6 * nb_; // complemented terms and one reduction round of a carry-save sum
        if (op.kind == RadixOp::Shl || op.kind == RadixOp::Shr) rows += 4 * nb_ + 8;
        if (op.kind == RadixOp::Div || op.kind == RadixOp::DivScalar) rows += 4 * 2 * nb_ + 5 * (nb_ + 1) + prop_rows(nb_ + 1);
    }
    return rows;
}

// One netlist level of integer operators (independent of each other), batched stage by stage.
void RadixEngine::run_level(helm_si_wires *w, const std::vector<RadixOp> &ops_in, int scratch)
{
    // a multiplication by a plaintext power of two IS a left shift: whole blocks move for free and an odd bit costs one
    // look-up per block, where the digit-wise product (block values up to 6) would need a full carry propagation
    // (x * 2: one round of bootstraps instead of six; same value mod 2^bits)
    std::vector<RadixOp> ops(ops_in);
    // x * s = x * (s mod 2^bits) mod 2^bits: a scalar is reduced to the operand's width FIRST, so that 2^s with s >= bits is
    // the product by zero it is and not a shift by s mod bits (helm_host_radix_level takes a raw 128-bit scalar)
    const unsigned __int128 width_mask = 2 * nb_ >= 128 ? ~(unsigned __int128)0 : (((unsigned __int128)1 << (2 * nb_)) - 1);
    for (auto &op : ops)
        if (op.kind == RadixOp::MulScalar) op.scalar &= width_mask;
    for (auto &op : ops)
        if (op.kind == RadixOp::MulScalar && op.scalar >= 2 && (op.scalar & (op.scalar - 1)) == 0) {
            int s = 0;
            while (!((op.scalar >> s) & 1)) s++;
            op.kind = RadixOp::ShlScalar;
            op.scalar = (unsigned __int128)s;
        }
    // ---- shifts by a plaintext amount first: their one round of look-ups rides in the level's first batch -------
    shift_scalar(w, ops);
    // ---- stage 1: block sums of add / sub, operand rows of the multiplications --------------
    struct Term { int base; int low; int maxv; }; // blocks below `low` are zero; block values <= maxv
    struct MulState { std::vector<Term> terms; int out; int out2 = -1; bool square = false; };
    std::vector<MulState> muls;
    std::vector<int32_t> prop_bases; // integers waiting for the final propagation
    int sp = scratch;
    auto take = [&](int rows) { const int b = sp; sp += rows; return b; };

    {
        std::vector<int32_t> li, lo;
        std::vector<int64_t> lc, ca;
        for (auto &op : ops) {
            switch (op.kind) {
            case RadixOp::Copy:
                for (int i = 0; i < nb_; i++) {
                    li.push_back(op.a + i); lc.push_back(1); li.push_back(-1); lc.push_back(0);
                    lo.push_back(op.out + i); ca.push_back(0);
                }
                break;
            case RadixOp::Add: case RadixOp::Sub: case RadixOp::AddScalar: case RadixOp::SubScalar: {
                if (op.a2 >= 0 || op.b2 >= 0 || op.out2 >= 0) break; // carry-save operands or result: through the term reduction below
                // a + b, or a + ~b + 1 with ~b digit = 3 - b_i (two's complement, mod 2^bits)
                const bool sub = op.kind == RadixOp::Sub || op.kind == RadixOp::SubScalar;
                const bool scalar = op.kind == RadixOp::AddScalar || op.kind == RadixOp::SubScalar;
                for (int i = 0; i < nb_; i++) {
                    int64_t c = 0;
                    li.push_back(op.a + i); lc.push_back(1);
                    if (scalar) {
                        const int digit = (int)((op.scalar >> (2 * i)) & 3);
                        li.push_back(-1); lc.push_back(0);
                        c = sub ? 3 - digit : digit;
                    } else {
                        li.push_back(op.b + i); lc.push_back(sub ? -1 : 1);
                        c = sub ? 3 : 0;
                    }
                    if (sub && i == 0) c += 1;
                    lo.push_back(op.out + i); ca.push_back(c);
                }
                prop_bases.push_back(op.out);
                break;
            }
            default:
                break; // multiplications, shifts, divisions: below
            }
        }
        lincomb(w, li, lc, ca, lo, 2);
    }

    // ---- multiplications: partial products ----------------------------------------------------
    {
        std::vector<int32_t> li, lo, in, lut, out;
        std::vector<int64_t> lc;
        std::vector<int32_t> zero_rows;
        for (auto &op : ops) {
            if (op.kind == RadixOp::MulScalar) {
                // sum_j s_j * (a << j blocks): block values <= 3 * s_j, no bootstrap needed here
                MulState ms;
                ms.out = op.out;
                for (int j = 0; j < nb_; j++) {
                    const int digit = (int)((op.scalar >> (2 * j)) & 3);
                    if (!digit) continue;
                    const int base = take(nb_);
                    for (int k = 0; k < nb_; k++) {
                        li.push_back(k >= j ? op.a + (k - j) : -1); lc.push_back(k >= j ? digit : 0);
                        li.push_back(-1); lc.push_back(0);
                        lo.push_back(base + k);
                    }
                    ms.terms.push_back(Term{base, j, 3 * digit});
                }
                muls.push_back(std::move(ms));
            } else if (op.kind == RadixOp::Mul) {
                // term L_j: block k = lo(a_{k-j} * b_j), term H_j: block k = hi(a_{k-1-j} * b_j)
                MulState ms;
                ms.out = op.out;
                ms.out2 = op.out2;
                if (op.a == op.b) {
                    // a square: a_j a_k and a_k a_j are the same product - one look-up on the pair gives lo / hi of
                    // 2 a_j a_k (<= 18: lo <= 3, hi <= 4).  Term L_j: block j + k = lo, term H_j: block j + k + 1 = hi,
                    // k >= j: (nb + 1) / 2 term pairs instead of nb, about half the look-ups, one reduction round fewer
                    ms.square = true;
                    for (int j = 0; 2 * j < nb_; j++) {
                        const int Lb = take(nb_), Hb = 2 * j + 1 < nb_ ? take(nb_) : -1;
                        for (int c = 0; c < nb_; c++) {
                            if (c >= 2 * j) { // packed operand 4 * a_k + a_j, k = c - j
                                li.push_back(op.a + (c - j)); lc.push_back(4);
                                li.push_back(op.a + j); lc.push_back(1);
                                lo.push_back(Lb + c);
                            } else
                                zero_rows.push_back(Lb + c);
                            if (Hb >= 0 && c <= 2 * j) zero_rows.push_back(Hb + c);
                        }
                        ms.terms.push_back(Term{Lb, 2 * j, 3});
                        if (Hb >= 0) ms.terms.push_back(Term{Hb, 2 * j + 1, 4});
                    }
                    muls.push_back(std::move(ms));
                    continue;
                }
                for (int j = 0; j < nb_; j++) {
                    const int Lb = take(nb_), Hb = j + 1 < nb_ ? take(nb_) : -1;
                    for (int k = 0; k < nb_; k++) {
                        if (k >= j) { // packed operand 4 * a_{k-j} + b_j, in place in the L row
                            li.push_back(op.a + (k - j)); lc.push_back(4);
                            li.push_back(op.b + j); lc.push_back(1);
                            lo.push_back(Lb + k);
                        } else
                            zero_rows.push_back(Lb + k);
                        if (Hb >= 0 && k <= j) zero_rows.push_back(Hb + k);
                    }
                    ms.terms.push_back(Term{Lb, j, 3});
                    if (Hb >= 0) ms.terms.push_back(Term{Hb, j + 1, 3});
                }
                muls.push_back(std::move(ms));
            }
        }
        lincomb(w, li, lc, {}, lo, 2);
        // bootstraps: hi first (reads the packed operand in the L row), then lo in place
        size_t mi = 0;
        std::vector<int32_t> in2, lut2, out2;
        for (auto &op : ops) {
            if (op.kind == RadixOp::MulScalar) { mi++; continue; }
            if (op.kind != RadixOp::Mul) continue;
            MulState &ms = muls[mi++];
            size_t ti = 0;
            if (ms.square) {
                for (int j = 0; 2 * j < nb_; j++) {
                    const int Lb = ms.terms[ti++].base, Hb = 2 * j + 1 < nb_ ? ms.terms[ti++].base : -1;
                    for (int c = 2 * j; c < nb_; c++) {
                        const bool diag = c == 2 * j;
                        if (Hb >= 0 && c + 1 < nb_) { in.push_back(Lb + c); lut.push_back(diag ? lut_mul_hi_ : lut_mul2_hi_); out.push_back(Hb + c + 1); }
                        in2.push_back(Lb + c); lut2.push_back(diag ? lut_mul_lo_ : lut_mul2_lo_); out2.push_back(Lb + c);
                    }
                }
                continue;
            }
            for (int j = 0; j < nb_; j++) {
                const int Lb = ms.terms[ti++].base, Hb = j + 1 < nb_ ? ms.terms[ti++].base : -1;
                for (int k = j; k < nb_; k++) {
                    if (Hb >= 0 && k + 1 < nb_) { in.push_back(Lb + k); lut.push_back(lut_mul_hi_); out.push_back(Hb + k + 1); }
                    in2.push_back(Lb + k); lut2.push_back(lut_mul_lo_); out2.push_back(Lb + k);
                }
            }
        }
        if (!zero_rows.empty()) {
            std::vector<uint64_t> z(zero_rows.size(), 0);
            auto guard = device_lock();
            si_ok(helm_si_wires_set_trivial(ctx_, w, zero_rows.data(), z.data(), (int64_t)zero_rows.size()), "set_trivial");
        }
        // one batch: the keyswitch of every ciphertext of a call finishes before any bootstrap writes,
        // so the lo look-ups may overwrite the packed rows the hi look-ups also read
        in.insert(in.end(), in2.begin(), in2.end());
        lut.insert(lut.end(), lut2.begin(), lut2.end());
        out.insert(out.end(), out2.begin(), out2.end());
        apply(w, in, lut, out);
    }

    // ---- additions / subtractions with an operand in carry-save form: their two to four terms join the reduction.
    //      x - y with y = y1 + y2: x + (~y1 + 1) + (~y2 + 1), ~t block = 3 - t block (mod 2^bits); the ones go to block 0
    //      of the first complemented term -------------------------------------------------------------------------
    {
        std::vector<int32_t> li, lo;
        std::vector<int64_t> lc, ca;
        for (auto &op : ops) {
            if (!((op.kind == RadixOp::Add || op.kind == RadixOp::Sub) && (op.a2 >= 0 || op.b2 >= 0 || op.out2 >= 0))) continue;
            MulState ms;
            ms.out = op.out;
            ms.out2 = op.out2;
            ms.terms.push_back(Term{op.a, 0, 3});
            if (op.a2 >= 0) ms.terms.push_back(Term{op.a2, 0, 3});
            if (op.kind == RadixOp::Add) {
                ms.terms.push_back(Term{op.b, 0, 3});
                if (op.b2 >= 0) ms.terms.push_back(Term{op.b2, 0, 3});
            } else {
                const int nc = op.b2 >= 0 ? 2 : 1;
                for (int t = 0; t < nc; t++) {
                    const int src = t == 0 ? op.b : op.b2, base = take(nb_);
                    for (int k = 0; k < nb_; k++) {
                        li.push_back(src + k); lc.push_back(-1);
                        li.push_back(-1); lc.push_back(0);
                        lo.push_back(base + k);
                        ca.push_back(3 + (t == 0 && k == 0 ? nc : 0));
                    }
                    ms.terms.push_back(Term{base, 0, t == 0 ? 3 + nc : 3});
                }
            }
            muls.push_back(std::move(ms));
        }
        lincomb(w, li, lc, ca, lo, 2);
    }
    // ---- multiplications: reduce the terms (sums of <= 15 per block, then message + carry) ----
    for (;;) {
        std::vector<int32_t> li, lo, in, lut, out, zero_rows;
        std::vector<int64_t> lc;
        bool any = false;
        const int T = 5;
        for (auto &ms : muls) {
            auto need_reduce = [&]() {
                if (ms.terms.size() > 2) return true;
                int s = 0;
                for (auto &t : ms.terms) s += t.maxv;
                return s > 6;
            };
            if (!need_reduce()) continue;
            any = true;
            std::sort(ms.terms.begin(), ms.terms.end(), [](const Term &x, const Term &y) { return x.low < y.low; });
            std::vector<Term> next;
            size_t q = 0;
            while (q < ms.terms.size()) {
                // greedy group: sum of the block bounds <= 15
                size_t e = q;
                int s = 0;
                while (e < ms.terms.size() && e - q < (size_t)T && s + ms.terms[e].maxv <= 15) s += ms.terms[e++].maxv;
                if (e - q == 1 && s <= 3) { // nothing to merge with: passes through
                    next.push_back(ms.terms[q]);
                    q = e;
                    continue;
                }
                const int low = ms.terms[q].low;
                const int Mb = take(nb_), Cb = low + 1 < nb_ ? take(nb_) : -1;
                for (int k = 0; k < nb_; k++) {
                    if (k < low) { zero_rows.push_back(Mb + k); if (Cb >= 0) zero_rows.push_back(Cb + k); continue; }
                    if (Cb >= 0 && k == low) zero_rows.push_back(Cb + k);
                    for (size_t u = q; u < q + (size_t)T; u++) {
                        const bool on = u < e && k >= ms.terms[u].low;
                        li.push_back(on ? ms.terms[u].base + k : -1);
                        lc.push_back(on ? 1 : 0);
                    }
                    lo.push_back(Mb + k);
                    if (Cb >= 0 && k + 1 < nb_) { in.push_back(Mb + k); lut.push_back(lut_carry_); out.push_back(Cb + k + 1); }
                }
                next.push_back(Term{Mb, low, 3});
                if (Cb >= 0) next.push_back(Term{Cb, low + 1, 3});
                q = e;
            }
            // message extraction after the carries have been read
            ms.terms.swap(next);
        }
        if (!any) break;
        lincomb(w, li, lc, {}, lo, T);
        if (!zero_rows.empty()) {
            std::vector<uint64_t> z(zero_rows.size(), 0);
            auto guard = device_lock();
            si_ok(helm_si_wires_set_trivial(ctx_, w, zero_rows.data(), z.data(), (int64_t)zero_rows.size()), "set_trivial");
        }
        // carries and (in place) messages of every summed row in one batch (see above)
        for (auto r : lo) { in.push_back(r); lut.push_back(lut_msg_); out.push_back(r); }
        apply(w, in, lut, out);
    }
    // ---- multiplications: last addition ---------------------------------------------------------
    {
        std::vector<int32_t> li, lo;
        std::vector<int64_t> lc;
        for (auto &ms : muls) {
            int s = 0;
            for (auto &t : ms.terms) s += t.maxv;
            if (ms.out2 >= 0) { // carry-save result: the two terms as they are, no propagation
                for (size_t u = 0; u < 2; u++)
                    for (int k = 0; k < nb_; k++) {
                        const bool on = u < ms.terms.size() && k >= ms.terms[u].low;
                        li.push_back(on ? ms.terms[u].base + k : -1);
                        lc.push_back(on ? 1 : 0);
                        li.push_back(-1);
                        lc.push_back(0);
                        lo.push_back((u == 0 ? ms.out : ms.out2) + k);
                    }
                continue;
            }
            for (int k = 0; k < nb_; k++) {
                for (size_t u = 0; u < 2; u++) {
                    const bool on = u < ms.terms.size() && k >= ms.terms[u].low;
                    li.push_back(on ? ms.terms[u].base + k : -1);
                    lc.push_back(on ? 1 : 0);
                }
                lo.push_back(ms.out + k);
            }
            if (s > 3) prop_bases.push_back(ms.out);
        }
        lincomb(w, li, lc, {}, lo, 2);
    }
    // ---- carry propagation of everything that needs it -------------------------------------------
    propagate(w, prop_bases, take(prop_rows(nb_) * (int)prop_bases.size()), nb_, nullptr);
    // ---- shifts by an encrypted amount and divisions (own round structure) ------------------------
    shift_encrypted(w, ops, sp);
    divide(w, ops, sp);
    apply(w, {}, {}, {}); // shift look-ups no batch has taken along
}

// ---------------------------------------------------------------------------------------
// ArithCircuit
// ---------------------------------------------------------------------------------------
static bool is_numeric_string(const std::string &s) // reference src/circuit.rs:100-102
{
    if (s.empty()) return false;
    for (char c : s)
        if (c < '0' || c > '9') return false;
    return true;
}

static int blocks_of(const std::string &ptxt_type)
{
    if (ptxt_type == "u8") return 4;
    if (ptxt_type == "u16") return 8;
    if (ptxt_type == "u32") return 16;
    if (ptxt_type == "u64") return 32;
    if (ptxt_type == "u128") return 64;
    throw Panic("internal error: entered unreachable code");
}

static PtxtType::Kind kind_of(const std::string &ptxt_type)
{
    if (ptxt_type == "u8") return PtxtType::U8;
    if (ptxt_type == "u16") return PtxtType::U16;
    if (ptxt_type == "u32") return PtxtType::U32;
    if (ptxt_type == "u64") return PtxtType::U64;
    return PtxtType::U128;
}

// connected components of the operator graph of a plan (wires produced by an operator connect it to its consumers;
// primary inputs and scalars connect nothing)
static size_t component_count(const std::vector<std::vector<RadixOp>> &plan)
{
    std::vector<int> parent;
    std::map<int, int> producer;
    std::vector<const RadixOp *> all;
    for (auto &lvl : plan)
        for (auto &op : lvl) {
            producer[op.out] = (int)all.size();
            parent.push_back((int)all.size());
            all.push_back(&op);
        }
    std::function<int(int)> find = [&](int x) { return parent[(size_t)x] == x ? x : parent[(size_t)x] = find(parent[(size_t)x]); };
    for (size_t id = 0; id < all.size(); id++)
        for (int row : {all[id]->a, all[id]->b}) {
            auto it = row >= 0 ? producer.find(row) : producer.end();
            if (it != producer.end()) parent[(size_t)find((int)id)] = find(it->second);
        }
    std::set<int> roots;
    for (size_t id = 0; id < all.size(); id++) roots.insert(find((int)id));
    return roots.size();
}

ArithCircuit::~ArithCircuit() = default;

ArithCircuit::ArithCircuit(helm_si_client_key *client_key, helm_si_ctx *server_key, Circuit circuit)
    : client_key_(client_key), server_key_(server_key), circuit_(std::move(circuit))
{
    si_ok(helm_si_get_params(server_key, &P_), "get_params");
}

void ArithCircuit::encrypt_value(SiEncWireMap &m, const std::string &wire, unsigned __int128 value)
{
    const int nb = m.blocks();
    std::vector<uint64_t> digits((size_t)nb);
    for (int i = 0; i < nb; i++) digits[(size_t)i] = (uint64_t)((value >> (2 * i)) & 3);
    std::vector<uint64_t> cts((size_t)nb * ((size_t)P_.k * P_.N + 1));
    if (helm_si_client_encrypt(client_key_, digits.data(), nb, cts.data())) throw Panic(client_key_ ? "encrypt failed" : "evaluation-only circuit (no client key): encrypt with the caller's keys and insert the ciphertext words");
    m.insert(wire, cts.data());
}

// reference src/circuit.rs:1113-1190
std::unique_ptr<SiEncWireMap> ArithCircuit::encrypt_inputs(const std::set<std::string> &wire_set,
                                                           const std::map<std::string, PtxtType> &input_wire_map)
{
    if (input_wire_map.empty()) throw Panic("called `Option::unwrap()` on a `None` value");
    switch (input_wire_map.begin()->second.kind) {
    case PtxtType::U8: global_ptxt_type_ = "u8"; break;
    case PtxtType::U16: global_ptxt_type_ = "u16"; break;
    case PtxtType::U32: global_ptxt_type_ = "u32"; break;
    case PtxtType::U64: global_ptxt_type_ = "u64"; break;
    case PtxtType::U128: global_ptxt_type_ = "u128"; break;
    default: throw Panic("internal error: entered unreachable code");
    }
    const int nb = blocks_of(global_ptxt_type_);
    auto m = std::make_unique<SiEncWireMap>(server_key_, nb);
    std::vector<std::string> names(wire_set.begin(), wire_set.end()); // FheType::None: rows exist, zero
    names.insert(names.end(), circuit_.input_wires().begin(), circuit_.input_wires().end());
    names.insert(names.end(), circuit_.dff_outputs().begin(), circuit_.dff_outputs().end());
    m->reserve_keys(names, 0);
    const bool dummy = input_wire_map.count("dummy") != 0;
    for (auto &input_wire : circuit_.input_wires()) {
        unsigned __int128 v = 0;
        if (!dummy) {
            auto it = input_wire_map.find(input_wire);
            if (it == input_wire_map.end()) throw Panic("\n Input wire \"" + input_wire + "\" not found in input wires!");
            v = it->second.value;
        }
        encrypt_value(*m, input_wire, v);
    }
    for (auto &w : circuit_.dff_outputs()) encrypt_value(*m, w, 0);
    return m;
}

// reference src/circuit.rs:1192-1216: trivial zeros
std::unique_ptr<SiEncWireMap> ArithCircuit::init_ready()
{
    const int nb = blocks_of(global_ptxt_type_.empty() ? "u32" : global_ptxt_type_);
    auto m = std::make_unique<SiEncWireMap>(server_key_, nb);
    m->reserve_keys(circuit_.output_wires(), 0);
    std::vector<int32_t> idx;
    for (auto &w : circuit_.output_wires())
        for (int b = 0; b < nb; b++) idx.push_back(m->row(w) + b);
    std::vector<uint64_t> zeros(idx.size(), 0);
    if (!idx.empty())
        si_ok(helm_si_wires_set_trivial(server_key_, m->table(), idx.data(), zeros.data(), (int64_t)idx.size()),
              "wires_set_trivial");
    return m;
}

// reference src/circuit.rs:1218-1297 multiplies by READY; arithmetic netlists carry no READY
// wire in any of the reference's runs or tests, so this mirrors the no-READY outcome.
void ArithCircuit::evaluate_ready(const SiEncWireMap &enc_wire_map, SiEncWireMap &)
{
    if (!enc_wire_map.contains_key("READY")) throw Panic("called `Option::unwrap()` on a `None` value (READY)");
    throw Panic("READY-latched outputs are not implemented for arithmetic circuits");
}

// reference src/circuit.rs:1299-1454
std::unique_ptr<SiEncWireMap> ArithCircuit::evaluate_encrypted(const SiEncWireMap &enc_wire_map, size_t cycle,
                                                               const std::string &ptxt_type)
{
    if (!circuit_.gates_empty()) throw Panic("assertion failed: self.circuit.gates.is_empty()");
    if (!circuit_.get_ordered_gates().empty()) throw Panic("assertion failed: self.circuit.ordered_gates.is_empty()");
    const int nb = blocks_of(ptxt_type);
    if (nb != enc_wire_map.blocks()) throw Panic("ptxt_type does not match the encrypted inputs");
    // Same-cycle memo, keyed on the cycle ALONE as in the reference (gates.rs:307-312 and every *_block method: `if
    // self.cycle == cycle { return cached }`, the operands are not looked at; tests/gates_test.rs:196-223).  Every gate
    // then hands back its cached output: the result is the given map with the gate outputs of that cycle, no launch.
    if (memo_on_ && memo_.hit(cycle) && memo_.out->blocks() == nb) {
        memo_hits_++;
        auto cached = enc_wire_map.clone(0);
        std::vector<int32_t> src, dst;
        for (auto &kv : circuit_.level_map())
            for (auto &g : kv.second) {
                const int32_t s0 = memo_.out->row(g.get_output_wire()), d0 = cached->row(g.get_output_wire());
                for (int b = 0; b < nb; b++) {
                    src.push_back(s0 + b);
                    dst.push_back(d0 + b);
                }
            }
        if (!src.empty())
            si_ok(helm_si_wires_copy(server_key_, memo_.out->table(), src.data(), cached->table(), dst.data(), (int64_t)src.size()),
                  "wires_copy");
        log_ += "  Cycle " + std::to_string(cycle) + " already evaluated: cached gate outputs returned\n";
        return cached;
    }
    auto remember = [&](SiEncWireMap &values) {
        if (!memo_on_) return;
        memo_.out = values.clone(0);
        memo_.cycle = cycle;
        memo_.in_id = enc_wire_map.id();
        memo_.in_gen = enc_wire_map.generation();
        memo_.valid = true;
    };
    const int bits = 2 * nb;
    const unsigned __int128 vmask = bits >= 128 ? ~(unsigned __int128)0 : (((unsigned __int128)1 << bits) - 1);
    RadixEngine eng(server_key_, nb);
    // plan every level first: the scratch region is sized once
    std::vector<std::vector<RadixOp>> plan;
    auto eval_values = enc_wire_map.clone(0);
    int64_t max_scratch = 0;
    for (auto &kv : circuit_.level_map()) {
        std::vector<RadixOp> ops;
        for (auto &g : kv.second) {
            const auto &ins = g.get_input_wires();
            RadixOp op{};
            op.out = eval_values->row(g.get_output_wire());
            bool is_ptxt_op = false;
            for (auto &w : ins) is_ptxt_op |= is_numeric_string(w);
            const GateType t = g.get_gate_type();
            if (is_ptxt_op) { // circuit.rs:1336-1387: ct (op) scalar, whatever the operand order
                op.a = -1;
                for (auto &w : ins) {
                    if (is_numeric_string(w)) {
                        unsigned __int128 x = 0;
                        bool overflow = false;
                        for (char c : w) {
                            x = x * 10 + (unsigned)(c - '0');
                            if (x > vmask) overflow = true;
                        }
                        op.scalar = overflow ? 0 : x; // parse::<uN>().unwrap_or(0)
                    } else
                        op.a = eval_values->row(w);
                }
                if (op.a < 0) throw Panic("Empty ctxt operand!");
                if (t == GateType::Add) op.kind = RadixOp::AddScalar;
                else if (t == GateType::Sub) op.kind = RadixOp::SubScalar;
                else if (t == GateType::Mult) op.kind = RadixOp::MulScalar;
                else if (t == GateType::Div) op.kind = RadixOp::DivScalar;
                else if (t == GateType::Shl) op.kind = RadixOp::ShlScalar;
                else if (t == GateType::Shr) op.kind = RadixOp::ShrScalar;
                else throw Panic("internal error: entered unreachable code");
            } else {
                if (ins.empty()) throw Panic("gate \"" + g.get_gate_name() + "\" has no input");
                op.a = eval_values->row(ins[0]);
                op.b = ins.size() > 1 ? eval_values->row(ins[1]) : -1;
                if (t == GateType::Copy) op.kind = RadixOp::Copy;
                else if (ins.size() < 2) throw Panic("index out of bounds: the len is 1 but the index is 1");
                else if (t == GateType::Add) op.kind = RadixOp::Add;
                else if (t == GateType::Sub) op.kind = RadixOp::Sub;
                else if (t == GateType::Div) op.kind = RadixOp::Div;
                else if (t == GateType::Shl) op.kind = RadixOp::Shl;
                else if (t == GateType::Shr) op.kind = RadixOp::Shr;
                else op.kind = RadixOp::Mul; // default arm (circuit.rs:1429-1435)
            }
            ops.push_back(op);
        }
        plan.push_back(std::move(ops));
    }
    // ---- carry-save planning: a product whose every consumer is an addition or a subtraction (possibly behind
    //      multiplications by powers of four = block shifts) keeps the two terms its reduction ends with instead of
    //      propagating carries (six rounds); the consumer sums the terms of both operands (one more reduction round)
    //      and propagates once.  a * b - c * d: 11 + 7 rounds in a row instead of 11 + 6 + ... 6.  The wire's own
    //      rows still receive the propagated value - in the level of the first adding consumer, where the rounds of
    //      that propagation ride in the consumer's batches.
    int64_t aux_total = 0;
    if (lazy_carries_) {
        const int aux_base = eval_values->scratch(0);
        auto block_shift = [&](const RadixOp &op) {
            if (op.kind != RadixOp::MulScalar || op.scalar < 4 || (op.scalar & (op.scalar - 1))) return false;
            int sh = 0;
            while (!((op.scalar >> sh) & 1)) sh++;
            return sh % 2 == 0;
        };
        std::set<int> output_rows;
        for (auto &wname : circuit_.output_wires())
            if (eval_values->contains_key(wname)) output_rows.insert(eval_values->row(wname));
        struct Use { size_t level, index; };
        std::map<int, std::vector<Use>> uses; // first row of a wire -> the operators that read it
        for (size_t l = 0; l < plan.size(); l++)
            for (size_t q = 0; q < plan[l].size(); q++)
                for (int row : {plan[l][q].a, plan[l][q].b})
                    if (row >= 0) uses[row].push_back({l, q});
        std::map<int, bool> lazy_ok; // the wire may be consumed in carry-save form
        for (size_t l = plan.size(); l-- > 0;)
            for (auto &op : plan[l]) {
                auto it = uses.find(op.out);
                bool ok = !output_rows.count(op.out) && it != uses.end();
                if (ok)
                    for (auto &u : it->second) {
                        const RadixOp &c = plan[u.level][u.index];
                        ok = ok && (c.kind == RadixOp::Add || c.kind == RadixOp::Sub || (block_shift(c) && lazy_ok[c.out]));
                    }
                lazy_ok[op.out] = ok;
            }
        struct Aux { int a, b, wire; };
        std::map<int, Aux> aux; // wire row -> rows of its two terms
        std::vector<std::vector<RadixOp>> finish(plan.size());
        for (size_t l = 0; l < plan.size(); l++)
            for (auto &op : plan[l]) {
                // operands first: terms of carry-save wires
                auto ia = op.a >= 0 ? aux.find(op.a) : aux.end(), ib = op.b >= 0 ? aux.find(op.b) : aux.end();
                const bool from_cs = ia != aux.end() && block_shift(op);
                // products, block shifts of carry-save wires, and (round 3) sums / differences themselves: a chain of
                // additions propagates carries once, at its end
                const bool make = lazy_ok[op.out] && (op.kind == RadixOp::Mul || from_cs || op.kind == RadixOp::Add || op.kind == RadixOp::Sub);
                if (make) {
                    const int base = aux_base + (int)aux_total;
                    aux_total += 2 * nb;
                    aux[op.out] = Aux{base, base + nb, op.out};
                }
                if (ia != aux.end() && (op.kind == RadixOp::Add || op.kind == RadixOp::Sub || from_cs)) {
                    op.a2 = ia->second.b;
                    op.a = ia->second.a;
                }
                if (ib != aux.end() && (op.kind == RadixOp::Add || op.kind == RadixOp::Sub)) {
                    op.b2 = ib->second.b;
                    op.b = ib->second.a;
                }
                if (make) {
                    const Aux &x = aux[op.out];
                    op.out = x.a;
                    op.out2 = x.b;
                }
            }
        // the propagated value of every carry-save wire, in the level of the first consumer that propagates carries
        // anyway (reached through block shifts and carry-save sums if need be)
        std::function<size_t(int)> first_adder = [&](int wire) -> size_t {
            size_t best = plan.size();
            for (auto &u : uses[wire]) {
                const RadixOp &c = plan[u.level][u.index];
                if ((c.kind == RadixOp::Add || c.kind == RadixOp::Sub) && c.out2 < 0) best = std::min(best, u.level);
                else if (c.out2 >= 0) // a block shift or a sum in carry-save form: the level where ITS terms are first added
                                      // up and propagated (its wire is the key of the entry holding these rows)
                    for (auto &kv : aux)
                        if (kv.second.a == c.out) best = std::min(best, first_adder(kv.first));
            }
            return best;
        };
        for (auto &kv : aux) {
            const size_t l = first_adder(kv.first);
            RadixOp f{};
            f.kind = RadixOp::Add;
            f.a = kv.second.a;
            f.b = kv.second.b;
            f.out = kv.second.wire;
            finish[std::min(l, plan.size() - 1)].push_back(f);
        }
        for (size_t l = 0; l < plan.size(); l++) plan[l].insert(plan[l].end(), finish[l].begin(), finish[l].end());
    }
    for (auto &ops : plan) max_scratch = std::max(max_scratch, eng.scratch_rows(ops));
    const size_t total_levels = circuit_.level_map().size();
    // Lanes by default: an operator graph with two or more connected components runs them concurrently on the server key
    // and ONE lane forked from it (identical ciphertexts, the rounds of the longest component in a row instead of the sum
    // over the levels).  add_lane() / clear_lanes() override: explicit lanes, or none.
    std::vector<helm_si_ctx *> lanes = lanes_;
    // Default (no explicit lanes): the components become chains on ONE context whose look-up rounds a RoundMerger joins
    // into launches of at most the device's capacity - no launch of one chain ever waits for compute units another
    // chain's launch holds (a bootstrap occupies its CU for the whole 8 ms, and a keyswitch + bootstrap pair that comes
    // 1 ms late loses two full rounds; measured: profiles/r03/arith_round_merger.txt).
    const bool merged = lanes.empty() && auto_lanes_ && helm_si_exchange_world(server_key_) <= 1 && component_count(plan) >= 2;
    if (!lanes.empty() || merged) {
        // ---- lanes: connected components of the operator graph (wires produced by an operator connect it to its
        //      consumers; primary inputs and scalars connect nothing), each component's levels compacted, components
        //      spread over the contexts by their bootstrap estimate, one host thread per context.
        std::vector<int> parent;
        std::vector<std::pair<size_t, size_t>> where; // op id -> (level, index)
        std::map<int, int> producer;                  // first row of an output -> op id
        for (size_t l = 0; l < plan.size(); l++)
            for (size_t q = 0; q < plan[l].size(); q++) {
                producer[plan[l][q].out] = (int)where.size();
                parent.push_back((int)where.size());
                where.push_back({l, q});
            }
        std::function<int(int)> find = [&](int x) { return parent[(size_t)x] == x ? x : parent[(size_t)x] = find(parent[(size_t)x]); };
        for (size_t id = 0; id < where.size(); id++) {
            const RadixOp &op = plan[where[id].first][where[id].second];
            for (int row : {op.a, op.b}) {
                auto it = row >= 0 ? producer.find(row) : producer.end();
                if (it != producer.end()) parent[(size_t)find((int)id)] = find(it->second);
            }
        }
        std::map<int, std::vector<int>> comps;
        for (size_t id = 0; id < where.size(); id++) comps[find((int)id)].push_back((int)id);
        auto cost = [&](const RadixOp &op) -> int64_t {
            switch (op.kind) {
            case RadixOp::Mul: return (int64_t)nb * nb * 2;
            case RadixOp::Div: case RadixOp::DivScalar: return (int64_t)nb * nb * 8;
            case RadixOp::Copy: return 0;
            default: return (int64_t)nb * 6;
            }
        };
        std::vector<std::pair<int64_t, int>> order; // (cost, root), heaviest first
        for (auto &c : comps) {
            int64_t w = 0;
            for (int id : c.second) w += cost(plan[where[(size_t)id].first][where[(size_t)id].second]);
            order.push_back({w, c.first});
        }
        std::sort(order.begin(), order.end(), [](auto &x, auto &y) { return x.first > y.first; });
        const size_t n_ctx = merged ? std::min<size_t>(comps.size(), 8) : 1 + lanes.size();
        auto ctx_of = [&](size_t lane) { return merged || lane == 0 ? server_key_ : lanes[lane - 1]; };
        std::vector<int64_t> load(n_ctx, 0);
        std::vector<std::vector<std::vector<RadixOp>>> lane_plan(n_ctx, std::vector<std::vector<RadixOp>>(plan.size()));
        for (auto &oc : order) {
            const size_t lane = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
            load[lane] += oc.first;
            for (int id : comps[oc.second]) lane_plan[lane][where[(size_t)id].first].push_back(plan[where[(size_t)id].first][where[(size_t)id].second]);
        }
        // the lane with the longest chain of bootstrap rounds is the critical path: its look-ups go first (merger: served
        // first in every launch; explicit lanes: helm_si_set_priority, dispatch priority of the context's own stream)
        std::unique_ptr<RoundMerger> merger;
        {
            auto rounds = [&](const RadixOp &op) -> int {
                int lg = 0;
                while ((1 << lg) < nb - 1) lg++;
                // propagate(): states + messages, the grouped carries (one round per level of groups of four, one more
                // per level beyond the first to turn c form into a bit), final messages
                std::function<int(int, bool)> carry_rounds = [&](int n, bool bit) {
                    return n <= 4 ? 1 : 1 + carry_rounds((n + 3) / 4, true) + (bit ? 1 : 0);
                };
                const int prop = nb <= 1 ? 1 : 2 + carry_rounds(nb, false);
                const bool cs = op.a2 >= 0 || op.b2 >= 0;
                switch (op.kind) {
                case RadixOp::Mul: return 5 + (op.out2 >= 0 ? 0 : prop);
                case RadixOp::Add: case RadixOp::Sub: return (op.out2 >= 0 ? 0 : prop) + (cs || op.out2 >= 0 ? 1 : 0);
                case RadixOp::AddScalar: case RadixOp::SubScalar: return prop;
                case RadixOp::MulScalar: return (op.scalar && !(op.scalar & (op.scalar - 1))) ? 1 : prop + 2;
                case RadixOp::ShlScalar: case RadixOp::ShrScalar: return 1;
                case RadixOp::Shl: case RadixOp::Shr: return 2 * (lg + 2);
                case RadixOp::Div: case RadixOp::DivScalar: return 2 * nb * (prop + 2);
                default: return 0;
                }
            };
            std::vector<int64_t> chain(n_ctx, 0);
            for (size_t lane = 0; lane < n_ctx; lane++)
                for (auto &ops : lane_plan[lane]) {
                    int m = 0;
                    for (auto &op : ops) m = std::max(m, rounds(op));
                    chain[lane] += m;
                }
            const size_t crit = (size_t)(std::max_element(chain.begin(), chain.end()) - chain.begin());
            if (merged) {
                int64_t capacity = round_capacity_;
                if (capacity <= 0) {
                    capacity = helm_si_round_capacity(server_key_);
                    if (capacity <= 0) // a failed occupancy query must not turn into one-ciphertext launches
                        throw Panic(std::string("helm_si_round_capacity: ") + helm_hip_last_error());
                }
                merger.reset(new RoundMerger(server_key_, (int)n_ctx, capacity, round_capacity_ > 0));
                for (size_t lane = 0; lane < n_ctx; lane++) merger->set_remaining((int)lane, chain[lane]);
            } else {
                for (size_t lane = 0; lane < n_ctx; lane++) si_ok(helm_si_set_priority(ctx_of(lane), lane == crit ? 1 : 0), "set_priority");
            }
        }
        std::vector<std::unique_ptr<RadixEngine>> engines;
        std::vector<int64_t> lane_scratch(n_ctx, 0);
        int64_t total_scratch = 0;
        for (size_t lane = 0; lane < n_ctx; lane++) {
            engines.emplace_back(new RadixEngine(ctx_of(lane), nb));
            if (merger) engines.back()->attach(merger.get(), (int)lane);
            for (auto &ops : lane_plan[lane]) lane_scratch[lane] = std::max(lane_scratch[lane], engines[lane]->scratch_rows(ops));
            total_scratch += lane_scratch[lane];
        }
        int base = eval_values->scratch(aux_total + total_scratch) + (int)aux_total; // carry-save terms first
        std::vector<int> lane_base(n_ctx);
        for (size_t lane = 0; lane < n_ctx; lane++) {
            lane_base[lane] = base;
            base += (int)lane_scratch[lane];
        }
        si_ok(helm_si_sync(server_key_), "sync"); // the inputs are in place before any lane reads them
        std::vector<std::string> errors(n_ctx);
        auto run_lane = [&](size_t lane) {
            try {
                for (auto &ops : lane_plan[lane])
                    if (!ops.empty()) engines[lane]->run_level(eval_values->table(), ops, lane_base[lane]);
                if (!merger) si_ok(helm_si_sync(ctx_of(lane)), "sync");
            } catch (const std::exception &e) {
                errors[lane] = e.what();
            }
            if (merger) merger->finish((int)lane);
        };
        std::vector<std::thread> threads;
        for (size_t lane = 1; lane < n_ctx; lane++) threads.emplace_back(run_lane, lane);
        run_lane(0);
        for (auto &t : threads) t.join();
        if (merger) si_ok(helm_si_sync(server_key_), "sync");
        for (auto &e : errors)
            if (!e.empty()) throw Panic(e);
        pbs_count_ = 0;
        pbs_rounds_ = 0;
        for (auto &e : engines) {
            pbs_count_ += e->pbs_count();
            pbs_rounds_ = std::max(pbs_rounds_, e->pbs_rounds()); // rounds in a row: the longest lane
        }
        if (merger) pbs_rounds_ = merger->launches(); // what the device ran one after the other
        std::ostringstream os;
        os << "  Evaluated " << comps.size() << " independent sub-circuit(s) of " << total_levels << " level(s) on " << n_ctx
           << (merger ? " chain(s), look-up rounds merged\n" : " lane(s)\n");
        log_ += os.str();
        remember(*eval_values);
        return eval_values;
    }
    const int scratch = eval_values->scratch(aux_total + max_scratch) + (int)aux_total; // carry-save terms first
    size_t li = 0;
    for (auto &kv : circuit_.level_map()) {
        eng.run_level(eval_values->table(), plan[li++], scratch);
        std::ostringstream os;
        os << "  Evaluated gates in level [" << kv.first << "/" << total_levels << "]\n";
        log_ += os.str();
    }
    si_ok(helm_si_sync(server_key_), "sync");
    pbs_count_ = eng.pbs_count();
    pbs_rounds_ = eng.pbs_rounds();
    remember(*eval_values);
    return eval_values;
}

// reference src/circuit.rs:1456-1483
std::map<std::string, PtxtType> ArithCircuit::decrypt_outputs(const SiEncWireMap &enc_wire_map, bool verbose)
{
    std::map<std::string, PtxtType> out;
    const int nb = enc_wire_map.blocks();
    const int bits = 2 * nb;
    const char *names[] = {"u8", "u16", "u32", "u64", "u128"};
    std::string pt = "u32";
    for (auto *n : names)
        if (blocks_of(n) == nb) pt = n;
    for (auto &w : circuit_.output_wires()) {
        auto ct = enc_wire_map.get(w);
        std::vector<uint64_t> vals((size_t)nb);
        if (helm_si_client_decrypt(client_key_, ct.data(), nb, vals.data())) throw Panic(client_key_ ? "decrypt failed" : "evaluation-only circuit (no client key): read the ciphertext words and decrypt with the caller's keys");
        unsigned __int128 v = 0;
        for (int i = nb - 1; i >= 0; i--) v = (v << 2) + vals[(size_t)i]; // carries included, wrapping
        if (bits < 128) v &= (((unsigned __int128)1 << bits) - 1);
        PtxtType p;
        p.kind = kind_of(pt);
        p.value = v;
        out[w] = p;
    }
    size_t i = 0;
    for (auto &kv : out) {
        if (i > 10 && !verbose) {
            log_ += "[!] More than ten output_wires, pass `--verbose` to see output.\n";
            break;
        }
        log_ += " " + kv.first + ": " + kv.second.to_string() + "\n";
        i++;
    }
    return out;
}

} // namespace helm
