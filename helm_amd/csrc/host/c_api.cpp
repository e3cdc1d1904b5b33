// c_api.cpp — flat C view (include/helm_host.h) of the C++ host front end.
#include "../../../include/helm_host.h"
#include "helm_host.hpp"
#include "../shard_rule.h"

#include <cstdlib>
#include <cstring>
#include <sstream>

using namespace helm;

struct helm_netlist { verilog_parser::Netlist nl; };
struct helm_circuit { Circuit c; };
struct helm_gate_circuit { std::unique_ptr<GateCircuit> gc; };
struct helm_enc_map { std::unique_ptr<EncWireMap> m; };
struct helm_si_enc_map { std::unique_ptr<SiEncWireMap> m; };
struct helm_si_circuit {
    std::unique_ptr<LutCircuit> lut;
    std::unique_ptr<ArithCircuit> arith;
    EvalCircuit<SiEncWireMap> *ec() { return lut ? static_cast<EvalCircuit<SiEncWireMap> *>(lut.get()) : arith.get(); }
};

namespace {
thread_local std::string g_err;
char *dup(const std::string &s)
{
    char *p = (char *)std::malloc(s.size() + 1);
    if (p) std::memcpy(p, s.c_str(), s.size() + 1);
    return p;
}
std::vector<std::string> lines(const char *text)
{
    std::vector<std::string> out;
    if (!text) return out;
    std::istringstream is(text);
    std::string l;
    while (std::getline(is, l))
        if (!l.empty()) out.push_back(l);
    return out;
}
const char *kind_name(PtxtType::Kind k)
{
    static const char *n[] = {"None", "Bool", "U8", "U16", "U32", "U64", "U128"};
    return n[(int)k];
}
std::string value_text(const PtxtType &v)
{
    if (v.kind == PtxtType::Bool) return v.value ? "1" : "0";
    if (v.kind == PtxtType::None) return "0";
    return v.to_string();
}
std::string map_text(const std::map<std::string, PtxtType> &m)
{
    std::ostringstream os;
    for (auto &kv : m) os << kv.first << '\t' << kind_name(kv.second.kind) << '\t' << value_text(kv.second) << '\n';
    return os.str();
}
std::map<std::string, PtxtType> parse_map(const char *text)
{
    std::map<std::string, PtxtType> m;
    for (auto &l : lines(text)) {
        const size_t a = l.find('\t'), b = l.find('\t', a + 1);
        if (a == std::string::npos || b == std::string::npos) throw Panic("malformed wire-map line: " + l);
        const std::string name = l.substr(0, a), kind = l.substr(a + 1, b - a - 1), val = l.substr(b + 1);
        PtxtType v;
        static const char *n[] = {"None", "Bool", "U8", "U16", "U32", "U64", "U128"};
        int k = -1;
        for (int i = 0; i < 7; i++)
            if (kind == n[i]) k = i;
        if (k < 0) throw Panic("unknown PtxtType kind: " + kind);
        v.kind = (PtxtType::Kind)k;
        unsigned __int128 x = 0;
        for (char ch : val) {
            if (ch < '0' || ch > '9') throw Panic("malformed value: " + val);
            x = x * 10 + (unsigned)(ch - '0');
        }
        v.value = x;
        m[name] = v;
    }
    return m;
}
std::string gate_line(const Gate &g, bool with_level)
{
    std::ostringstream os;
    os << g.get_gate_name() << '\t' << gate_type_name(g.get_gate_type()) << '\t' << g.get_output_wire() << '\t';
    if (g.get_lut_const()) {
        for (auto b : *g.get_lut_const()) os << b;
    } else
        os << '-';
    os << '\t';
    for (size_t i = 0; i < g.get_input_wires().size(); i++) os << (i ? "," : "") << g.get_input_wires()[i];
    if (with_level) os << '\t' << g.get_level();
    os << '\n';
    return os.str();
}
template <typename F> int guard(F f)
{
    try {
        f();
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}
std::set<std::string> to_set(const char *t)
{
    auto v = lines(t);
    return std::set<std::string>(v.begin(), v.end());
}
} // namespace

extern "C" {

const char *helm_host_last_error(void) { return g_err.c_str(); }
void helm_host_free(char *text) { std::free(text); }

int helm_host_read_verilog_file(const char *file_name, int is_arith, helm_netlist **out)
{
    return guard([&] { *out = new helm_netlist{verilog_parser::read_verilog_file(file_name, is_arith != 0)}; });
}
int helm_host_read_verilog_text(const char *text, int is_arith, helm_netlist **out)
{
    return guard([&] { *out = new helm_netlist{verilog_parser::read_verilog_text(text, is_arith != 0)}; });
}
void helm_host_netlist_free(helm_netlist *nl) { delete nl; }

char *helm_host_netlist_list(const helm_netlist *nl, int which)
{
    std::ostringstream os;
    switch (which) {
    case 0: for (auto &kv : nl->nl.gates) os << gate_line(kv.second, false); break;
    case 1: for (auto &w : nl->nl.wire_set) os << w << '\n'; break;
    case 2: for (auto &w : nl->nl.inputs) os << w << '\n'; break;
    case 3: for (auto &w : nl->nl.outputs) os << w << '\n'; break;
    case 4: for (auto &w : nl->nl.dff_outputs) os << w << '\n'; break;
    default: break;
    }
    return dup(os.str());
}
int helm_host_netlist_flags(const helm_netlist *nl, int *has_luts, int *has_arith)
{
    *has_luts = nl->nl.has_luts;
    *has_arith = nl->nl.has_arith;
    return 0;
}
int helm_host_read_input_wires(const char *file_name, const char *ptxt_type, char **out_map)
{
    return guard([&] { *out_map = dup(map_text(verilog_parser::read_input_wires(file_name, ptxt_type))); });
}
int helm_host_write_output_wires(const char *file_name, const char *wire_map)
{
    return guard([&] { verilog_parser::write_output_wires(std::string(file_name), parse_map(wire_map)); });
}
int helm_host_parse_input_wire(const char *wire, const char *ptxt_type, char **out_value)
{
    return guard([&] {
        std::map<std::string, PtxtType> m{{"v", parse_input_wire(wire, ptxt_type)}};
        *out_value = dup(map_text(m));
    });
}
int helm_host_hex_to_bitstring(const char *hex, char **out_bits)
{
    return guard([&] { *out_bits = dup(hex_to_bitstring(hex)); });
}

int helm_host_circuit_new(const helm_netlist *gates_from, const char *input_wires, const char *output_wires,
                          const char *dff_outputs, helm_circuit **out)
{
    return guard([&] {
        *out = new helm_circuit{Circuit(gates_from->nl.gates, lines(input_wires), lines(output_wires), lines(dff_outputs))};
    });
}
void helm_host_circuit_free(helm_circuit *c) { delete c; }
int helm_host_circuit_sort_circuit(helm_circuit *c) { return guard([&] { c->c.sort_circuit(); }); }
int helm_host_circuit_compute_levels(helm_circuit *c) { return guard([&] { c->c.compute_levels(); }); }
char *helm_host_circuit_get_ordered_gates(const helm_circuit *c)
{
    std::ostringstream os;
    for (auto &g : c->c.get_ordered_gates()) os << gate_line(g, true);
    return dup(os.str());
}
char *helm_host_circuit_level_map(const helm_circuit *c)
{
    std::ostringstream os;
    for (auto &kv : c->c.level_map())
        for (auto &g : kv.second) os << gate_line(g, true);
    return dup(os.str());
}
int helm_host_circuit_initialize_wire_map(const helm_circuit *c, const char *wire_set, const char *user_inputs,
                                          const char *ptxt_type, char **out_map)
{
    return guard([&] { *out_map = dup(map_text(c->c.initialize_wire_map(to_set(wire_set), parse_map(user_inputs), ptxt_type))); });
}
int helm_host_circuit_evaluate(helm_circuit *c, const char *wire_map, char **out_map)
{
    return guard([&] { *out_map = dup(map_text(c->c.evaluate(parse_map(wire_map)))); });
}

int helm_host_preprocess(const char *text, int arithmetic, char **out)
{
    if (!text || !out) {
        g_err = "null argument";
        return -1;
    }
    try {
        *out = dup(preprocess(text, arithmetic != 0));
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

int helm_host_pack_levels(const int32_t *opcode, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                          const int32_t *out, const int64_t *level_offsets, int64_t n_levels, int64_t quantum,
                          int64_t *order, int64_t *new_offsets, int64_t *n_launches)
{
    if (!opcode || !in0 || !in1 || !in2 || !out || !level_offsets || !order || !new_offsets || !n_launches || n_levels < 0) {
        g_err = "null argument";
        return -1;
    }
    try {
        std::vector<int64_t> ord, off;
        const int rc = pack_levels(opcode, in0, in1, in2, out, level_offsets, n_levels, quantum, ord, off);
        std::copy(ord.begin(), ord.end(), order);
        std::copy(off.begin(), off.end(), new_offsets);
        *n_launches = (int64_t)off.size() - 1;
        return rc;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

int helm_host_pack_levels_costed(const int32_t *opcode, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                                 const int32_t *out, const int64_t *level_offsets, int64_t n_levels, int64_t quantum,
                                 const double *quarter_cost, int64_t *order, int64_t *new_offsets, int64_t *n_launches)
{
    if (!opcode || !in0 || !in1 || !in2 || !out || !level_offsets || !order || !new_offsets || !n_launches || n_levels < 0) {
        g_err = "null argument";
        return -1;
    }
    if (quarter_cost)
        for (int q = 0; q < 4; q++)
            if (!(quarter_cost[q] > 0.0)) {
                g_err = "pack_levels: quarter costs must be positive";
                return -1;
            }
    try {
        std::vector<int64_t> ord, off;
        const int rc = pack_levels(opcode, in0, in1, in2, out, level_offsets, n_levels, quantum, ord, off, quarter_cost);
        std::copy(ord.begin(), ord.end(), order);
        std::copy(off.begin(), off.end(), new_offsets);
        *n_launches = (int64_t)off.size() - 1;
        return rc;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

int64_t helm_host_shard_bounds(const int32_t *opcode, int64_t count, int world, int64_t *bounds)
{
    if ((!opcode && count > 0) || !bounds || count < 0 || world <= 0) {
        g_err = "shard_bounds: bad argument";
        return -1;
    }
    return helm_shard::chunk_bounds(opcode, count, world, bounds);
}

int helm_host_enc_map_new(helm_hip_ctx *server_key, helm_enc_map **out)
{
    return guard([&] {
        helm_hip_params P;
        if (!server_key || helm_hip_get_params(server_key, &P)) throw Panic("null server key");
        *out = new helm_enc_map{std::make_unique<EncWireMap>(server_key, P.n)};
    });
}

int helm_host_gate_circuit_new(helm_client_key *client_key, helm_hip_ctx *server_key, const helm_circuit *circuit,
                               helm_gate_circuit **out)
{
    return guard([&] {
        if (!server_key || !circuit) throw Panic("null argument"); // client_key = NULL: evaluation only (see helm_host.h)
        *out = new helm_gate_circuit{std::make_unique<GateCircuit>(client_key, server_key, circuit->c)};
    });
}
void helm_host_gate_circuit_free(helm_gate_circuit *gc) { delete gc; }

int helm_host_gate_circuit_encrypt_inputs(helm_gate_circuit *gc, const char *wire_set, const char *input_wire_map,
                                          helm_enc_map **out)
{
    return guard([&] { *out = new helm_enc_map{gc->gc->encrypt_inputs(to_set(wire_set), parse_map(input_wire_map))}; });
}
int helm_host_gate_circuit_evaluate_encrypted(helm_gate_circuit *gc, const helm_enc_map *enc_wire_map,
                                              int64_t current_cycle, const char *ptxt_type, helm_enc_map **out)
{
    return guard([&] {
        *out = new helm_enc_map{gc->gc->evaluate_encrypted(*enc_wire_map->m, (size_t)current_cycle, ptxt_type ? ptxt_type : "bool")};
    });
}
int helm_host_gate_circuit_init_ready(helm_gate_circuit *gc, helm_enc_map **out)
{
    return guard([&] { *out = new helm_enc_map{gc->gc->init_ready()}; });
}
int helm_host_gate_circuit_evaluate_ready(helm_gate_circuit *gc, const helm_enc_map *enc_wire_map, helm_enc_map *valid_outputs)
{
    return guard([&] { gc->gc->evaluate_ready(*enc_wire_map->m, *valid_outputs->m); });
}
int helm_host_gate_circuit_decrypt_outputs(helm_gate_circuit *gc, const helm_enc_map *enc_wire_map, int verbose, char **out_map)
{
    return guard([&] { *out_map = dup(map_text(gc->gc->decrypt_outputs(*enc_wire_map->m, verbose != 0))); });
}
char *helm_host_gate_circuit_log(helm_gate_circuit *gc) { return dup(gc->gc->log()); }
int64_t helm_host_gate_circuit_pbs_per_cycle(const helm_gate_circuit *gc) { return gc->gc->pbs_per_cycle(); }
int helm_host_gate_circuit_shard_over(helm_gate_circuit *gc, helm_comm *comm, int64_t replicate_below)
{
    return guard([&] { gc->gc->shard_over(comm, replicate_below); });
}
int helm_host_gate_circuit_set_exchange_overlap(helm_gate_circuit *gc, int on)
{
    return guard([&] { gc->gc->set_exchange_overlap(on != 0); });
}
int64_t helm_host_gate_circuit_memo_hits(const helm_gate_circuit *gc) { return gc->gc->memo_hits(); }

void helm_host_enc_map_free(helm_enc_map *m) { delete m; }
int helm_host_enc_map_insert(helm_enc_map *m, const char *wire, const uint32_t *lwe)
{
    return guard([&] { m->m->insert(wire, lwe); });
}
int helm_host_enc_map_get(const helm_enc_map *m, const char *wire, uint32_t *lwe_out)
{
    return guard([&] {
        auto v = m->m->get(wire);
        std::memcpy(lwe_out, v.data(), v.size() * sizeof(uint32_t));
    });
}
int helm_host_enc_map_contains_key(const helm_enc_map *m, const char *wire) { return m->m->contains_key(wire) ? 1 : 0; }
char *helm_host_enc_map_keys(const helm_enc_map *m)
{
    std::ostringstream os;
    for (auto &k : m->m->keys()) os << k << '\n';
    return dup(os.str());
}


int helm_host_si_circuit_new(int mode, helm_si_client_key *client_key, helm_si_ctx *server_key,
                             const helm_circuit *circuit, helm_si_circuit **out)
{
    return guard([&] {
        // client_key = NULL: an evaluation-only circuit - the caller encrypts and decrypts with its own keys (tfhe's, in the
        // Rust shim) and moves ciphertext words through helm_host_si_enc_map_insert / _get
        if (!server_key || !circuit) throw Panic("null argument");
        auto *c = new helm_si_circuit();
        try {
            if (mode == 0) c->lut = std::make_unique<LutCircuit>(client_key, server_key, circuit->c);
            else c->arith = std::make_unique<ArithCircuit>(client_key, server_key, circuit->c);
        } catch (...) {
            delete c;
            throw;
        }
        *out = c;
    });
}
void helm_host_si_circuit_free(helm_si_circuit *c) { delete c; }
int helm_host_si_circuit_encrypt_inputs(helm_si_circuit *c, const char *wire_set, const char *input_wire_map,
                                        helm_si_enc_map **out)
{
    return guard([&] { *out = new helm_si_enc_map{c->ec()->encrypt_inputs(to_set(wire_set), parse_map(input_wire_map))}; });
}
int helm_host_si_circuit_evaluate_encrypted(helm_si_circuit *c, const helm_si_enc_map *enc_wire_map,
                                            int64_t current_cycle, const char *ptxt_type, helm_si_enc_map **out)
{
    return guard([&] {
        *out = new helm_si_enc_map{c->ec()->evaluate_encrypted(*enc_wire_map->m, (size_t)current_cycle, ptxt_type ? ptxt_type : "bool")};
    });
}
int helm_host_si_circuit_init_ready(helm_si_circuit *c, helm_si_enc_map **out)
{
    return guard([&] { *out = new helm_si_enc_map{c->ec()->init_ready()}; });
}
int helm_host_si_circuit_evaluate_ready(helm_si_circuit *c, const helm_si_enc_map *enc_wire_map, helm_si_enc_map *valid_outputs)
{
    return guard([&] { c->ec()->evaluate_ready(*enc_wire_map->m, *valid_outputs->m); });
}
int helm_host_si_circuit_decrypt_outputs(helm_si_circuit *c, const helm_si_enc_map *enc_wire_map, int verbose, char **out_map)
{
    return guard([&] { *out_map = dup(map_text(c->ec()->decrypt_outputs(*enc_wire_map->m, verbose != 0))); });
}
int helm_host_si_circuit_set_wopbs(helm_si_circuit *c, helm_wop_ctx *wop, int bits_per_block)
{
    if (!c || !c->lut) {
        g_err = "set_wopbs: not a LUT-mode circuit";
        return -1;
    }
    return guard([&] { c->lut->set_wide_lut_key(wop, bits_per_block); });
}
int helm_host_si_circuit_add_lane(helm_si_circuit *c, helm_si_ctx *lane)
{
    if (!c || !c->arith) {
        g_err = "add_lane: not an arithmetic-mode circuit";
        return -1;
    }
    return guard([&] {
        if (lane) c->arith->add_lane(lane);
        else c->arith->clear_lanes();
    });
}
// FheUintN operators of one level (gates.rs:306-702) as one batched call: what a Rust HipArithCircuit forwards to
static std::vector<RadixOp> to_radix_ops(const helm_radix_op *ops, int64_t count)
{
    if (!ops || count < 0) throw Panic("radix level: bad argument");
    std::vector<RadixOp> v((size_t)count);
    for (int64_t g = 0; g < count; g++) {
        if (ops[g].kind < 0 || ops[g].kind > HELM_RADIX_SHR_SCALAR) throw Panic("radix level: unknown operator kind");
        v[(size_t)g].kind = (RadixOp::Kind)ops[g].kind;
        v[(size_t)g].a = ops[g].a;
        v[(size_t)g].b = ops[g].b;
        v[(size_t)g].out = ops[g].out;
        v[(size_t)g].scalar = ((unsigned __int128)ops[g].scalar_hi << 64) | ops[g].scalar_lo;
    }
    return v;
}
int64_t helm_host_radix_scratch_rows(helm_si_ctx *ctx, int32_t blocks, const helm_radix_op *ops, int64_t count)
{
    int64_t rows = -1;
    guard([&] {
        if (!ctx || blocks < 1) throw Panic("radix level: bad argument");
        RadixEngine eng(ctx, blocks);
        rows = eng.scratch_rows(to_radix_ops(ops, count));
    });
    return rows;
}
int helm_host_radix_level(helm_si_ctx *ctx, helm_si_wires *wires, int32_t blocks, const helm_radix_op *ops, int64_t count,
                          int32_t scratch_first_row, int64_t *pbs_out, int64_t *rounds_out)
{
    return guard([&] {
        if (!ctx || !wires || blocks < 1 || scratch_first_row < 0) throw Panic("radix level: bad argument");
        RadixEngine eng(ctx, blocks);
        eng.run_level(wires, to_radix_ops(ops, count), scratch_first_row);
        if (pbs_out) *pbs_out = eng.pbs_count();
        if (rounds_out) *rounds_out = eng.pbs_rounds();
    });
}
int helm_host_si_circuit_set_lazy_carries(helm_si_circuit *c, int on)
{
    if (!c || !c->arith) {
        g_err = "set_lazy_carries: not an arithmetic-mode circuit";
        return -1;
    }
    c->arith->set_lazy_carries(on != 0);
    return 0;
}
int helm_host_si_circuit_set_round_capacity(helm_si_circuit *c, int64_t capacity)
{
    if (!c || !c->arith || capacity < 0) {
        g_err = "set_round_capacity: not an arithmetic-mode circuit, or a negative capacity";
        return -1;
    }
    c->arith->set_round_capacity(capacity);
    return 0;
}
int helm_host_si_circuit_set_memo(helm_si_circuit *c, int on)
{
    if (!c || !c->arith) {
        g_err = "set_memo: not an arithmetic-mode circuit";
        return -1;
    }
    c->arith->set_memo(on != 0);
    return 0;
}
int helm_host_si_circuit_reset_memo(helm_si_circuit *c)
{
    if (!c || !c->arith) {
        g_err = "reset_memo: not an arithmetic-mode circuit";
        return -1;
    }
    c->arith->reset_memo();
    return 0;
}
int helm_host_si_circuit_set_timing_lines(helm_si_circuit *c, int on)
{
    if (!c || !c->lut) {
        g_err = "set_timing_lines: not a LUT-mode circuit";
        return -1;
    }
    c->lut->set_timing_lines(on != 0);
    return 0;
}
char *helm_host_si_circuit_log(helm_si_circuit *c) { return dup(c->lut ? c->lut->log() : c->arith->log()); }
int64_t helm_host_si_circuit_pbs_per_cycle(const helm_si_circuit *c)
{
    return c->lut ? c->lut->pbs_per_cycle() : c->arith->pbs_per_cycle();
}
int64_t helm_host_si_circuit_pbs_rounds_per_cycle(const helm_si_circuit *c)
{
    return c->lut ? 0 : c->arith->pbs_rounds_per_cycle();
}
int64_t helm_host_si_circuit_memo_hits(const helm_si_circuit *c) { return c->lut ? c->lut->memo_hits() : c->arith->memo_hits(); }
int helm_host_si_enc_map_new(helm_si_ctx *server_key, int blocks, helm_si_enc_map **out)
{
    return guard([&] {
        if (!server_key || blocks < 1) throw Panic("bad argument");
        *out = new helm_si_enc_map{std::make_unique<SiEncWireMap>(server_key, blocks)};
    });
}
void helm_host_si_enc_map_free(helm_si_enc_map *m) { delete m; }
int helm_host_si_enc_map_blocks(const helm_si_enc_map *m) { return m->m->blocks(); }
int helm_host_si_enc_map_row_words(const helm_si_enc_map *m) { return m->m->row_words(); }
int helm_host_si_enc_map_insert(helm_si_enc_map *m, const char *wire, const uint64_t *lwe)
{
    return guard([&] { m->m->insert(wire, lwe); });
}
int helm_host_si_enc_map_get(const helm_si_enc_map *m, const char *wire, uint64_t *lwe_out)
{
    return guard([&] {
        auto v = m->m->get(wire);
        std::memcpy(lwe_out, v.data(), v.size() * sizeof(uint64_t));
    });
}
int helm_host_si_enc_map_contains_key(const helm_si_enc_map *m, const char *wire) { return m->m->contains_key(wire) ? 1 : 0; }
char *helm_host_si_enc_map_keys(const helm_si_enc_map *m)
{
    std::ostringstream os;
    for (auto &k : m->m->keys()) os << k << '\n';
    return dup(os.str());
}

} // extern "C"
