// helm_host.hpp — host side of the gates-mode path, C++ mirror of the reference's
// Rust interface (the Rust toolchain is absent from this image):
//
//   helm::GateType, helm::Gate            reference src/gates.rs:23-60, 104-280
//   helm::verilog_parser::*               reference src/verilog_parser.rs
//   helm::Circuit                         reference src/circuit.rs:60-67, 104-381
//   helm::EvalCircuit / helm::GateCircuit reference src/circuit.rs:35-58, 449-577
//   helm::PtxtType, parse_input_wire,
//   get_input_wire_map, hex_to_bitstring  reference src/lib.rs:20-29, 90-194
//
// Same names, argument meaning and error behaviour; where the reference panics this
// code throws helm::Panic carrying the same message.
#pragma once
#include <cstdint>
#include <condition_variable>
#include <map>
#include <mutex>
#include <memory>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/helm_client.h"
#include "../../../include/helm_wopbs.h"
#include "../../../include/helm_hip.h"
#include "../../../include/helm_comm.h"
#include "../../../include/helm_shortint.h"
#include "../../../include/helm_wopbs.h"

namespace helm {

struct Panic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// reference src/lib.rs:20-29
struct PtxtType {
    enum Kind { None = 0, Bool, U8, U16, U32, U64, U128 } kind = None;
    unsigned __int128 value = 0;
    static PtxtType none() { return {}; }
    static PtxtType boolean(bool b) { return {Bool, (unsigned __int128)(b ? 1 : 0)}; }
    bool as_bool() const { return value != 0; }
    bool operator==(const PtxtType &o) const { return kind == o.kind && value == o.value; }
    bool operator!=(const PtxtType &o) const { return !(*this == o); }
    std::string to_string() const; // Rust Display of the inner value ("true"/"false"/integer)
};

// reference src/gates.rs:23-45 (declaration order = helm_gate_op)
enum class GateType : int {
    And = 0, Dff, Lut, Mux, Nand, Nor, Not, Or, Xnor, Xor, Buf, ConstOne, ConstZero,
    Mult, Add, Sub, Div, Shl, Shr, Copy
};
const char *gate_type_name(GateType t);

// reference src/gates.rs:47-60, 104-149
class Gate {
  public:
    Gate(std::string gate_name, GateType gate_type, std::vector<std::string> input_wires,
         std::optional<std::vector<uint64_t>> lut_const, std::string output_wire, size_t level)
        : gate_name_(std::move(gate_name)), gate_type_(gate_type), input_wires_(std::move(input_wires)),
          lut_const_(std::move(lut_const)), output_wire_(std::move(output_wire)), level_(level)
    {
    }
    const std::vector<std::string> &get_input_wires() const { return input_wires_; }
    const std::string &get_output_wire() const { return output_wire_; }
    GateType get_gate_type() const { return gate_type_; }
    const std::string &get_gate_name() const { return gate_name_; }
    const std::optional<std::vector<uint64_t>> &get_lut_const() const { return lut_const_; }
    void set_level(size_t level) { level_ = level; }
    size_t get_level() const { return level_; }
    // reference src/gates.rs:151-239
    PtxtType evaluate(const std::vector<PtxtType> &input_values);
    // identity / ordering by gate name, reference src/gates.rs:62-87
    bool operator==(const Gate &o) const { return gate_name_ == o.gate_name_; }
    bool operator<(const Gate &o) const { return gate_name_ < o.gate_name_; }
    std::string debug() const; // reference src/gates.rs:89-102

  private:
    std::string gate_name_;
    GateType gate_type_;
    std::vector<std::string> input_wires_;
    std::optional<std::vector<uint64_t>> lut_const_;
    std::string output_wire_;
    size_t level_;
    size_t cycle_ = 0;
    PtxtType output_;
};

// HashSet<Gate> keyed by gate name (a second gate with the same name is dropped,
// like HashSet::insert).
using GateSet = std::map<std::string, Gate>;

namespace verilog_parser {
struct Netlist {
    GateSet gates;
    std::set<std::string> wire_set; // gate-output wires
    std::vector<std::string> inputs, outputs, dff_outputs;
    bool has_luts = false, has_arith = false;
};
Gate parse_gate(const std::vector<std::string> &tokens);                 // verilog_parser.rs:31-120
std::optional<std::pair<size_t, size_t>> parse_range(const std::string &s); // :122-136
Netlist read_verilog_file(const std::string &file_name, bool is_arith); // :138-276
Netlist read_verilog_text(const std::string &text, bool is_arith);      // same, from memory
std::map<std::string, PtxtType> read_input_wires(const std::string &file_name, const std::string &ptxt_type); // :278-317
void write_output_wires(const std::optional<std::string> &file_name,
                        const std::map<std::string, PtxtType> &input_map); // :319-349
} // namespace verilog_parser

PtxtType parse_input_wire(const std::string &wire, const std::string &ptxt_type); // lib.rs:90-106
std::string hex_to_bitstring(const std::string &hex);                             // lib.rs:181-194
std::map<std::string, PtxtType> get_input_wire_map(const std::optional<std::string> &inputs_filename,
                                                   const std::vector<std::vector<std::string>> &wire_inputs,
                                                   const std::string &arithmetic_type); // lib.rs:113-179

// reference src/circuit.rs:60-67, 104-381
class Circuit {
  public:
    Circuit(GateSet gates, std::vector<std::string> input_wires, std::vector<std::string> output_wires,
            std::vector<std::string> dff_outputs)
        : gates_(std::move(gates)), input_wires_(std::move(input_wires)), output_wires_(std::move(output_wires)),
          dff_outputs_(std::move(dff_outputs))
    {
    }
    void sort_circuit();   // circuit.rs:122-171
    void compute_levels(); // circuit.rs:174-239
    std::map<std::string, PtxtType> initialize_wire_map(const std::set<std::string> &wire_set,
                                                        const std::map<std::string, PtxtType> &user_inputs,
                                                        const std::string &ptxt_type) const; // :245-333
    std::string print_level_map() const;                                                       // :335-342
    const std::vector<Gate> &get_ordered_gates() const { return ordered_gates_; }
    std::map<std::string, PtxtType> evaluate(const std::map<std::string, PtxtType> &wire_map); // :348-381

    const std::map<size_t, std::vector<Gate>> &level_map() const { return level_map_; }
    const std::vector<std::string> &input_wires() const { return input_wires_; }
    const std::vector<std::string> &output_wires() const { return output_wires_; }
    const std::vector<std::string> &dff_outputs() const { return dff_outputs_; }
    bool gates_empty() const { return gates_.empty(); }

  private:
    GateSet gates_;
    std::vector<std::string> input_wires_, output_wires_, dff_outputs_;
    std::vector<Gate> ordered_gates_;
    std::map<size_t, std::vector<Gate>> level_map_;
};

// Raw synthesis output (Yosys structural Verilog; with `arithmetic`, behavioural assign statements) -> the
// netlist dialect read_verilog_file accepts (preprocessor.cpp; reference README.md:116-120,133-137).
std::string preprocess(const std::string &text, bool arithmetic);

// Launch packing (level_pack.cpp): the level map of circuit.rs:174-239 as index arrays -> `order`
// (new position -> gate) and `new_off` (launch boundaries) such that every launch but the last few
// holds a whole number of `quantum` bootstraps.  Returns 0, or 1 when the schedule was kept as it is.
// quarter_cost (optional, 4 entries: helm_hip_launch_costs()): launches narrower than a round take the width with the best
// bootstraps-per-cost and leave the rest to the next launch.
int pack_levels(const int32_t *op, const int32_t *in0, const int32_t *in1, const int32_t *in2, const int32_t *out,
                const int64_t *off, int64_t n_levels, int64_t quantum, std::vector<int64_t> &order,
                std::vector<int64_t> &new_off, const double *quarter_cost = nullptr);

uint64_t next_map_id(); // process-wide counter behind EncWireMap::id() / SiEncWireMap::id()

// Same-cycle memo of an evaluator (reference src/gates.rs:55-59: every Gate keeps `cycle` and its last encrypted
// output, and gates.rs:288-292, 307-312 return it when called again in that cycle).  All gates of a circuit are
// evaluated with the same cycle, so the memo lives once per circuit: the wire map the cycle produced.
// The evaluators' progress text (what the reference prints) is kept until log() is read; a host that never reads it must
// not grow without bound: beyond 1 MiB the older half is dropped.
inline void append_log(std::string &log, const std::string &text)
{
    log += text;
    if (log.size() > (1u << 20)) log = "[... older lines dropped ...]\n" + log.substr(log.size() - (1u << 19));
}

template <typename MapT> struct CycleMemo {
    bool valid = false;
    size_t cycle = 0;
    uint64_t in_id = 0, in_gen = 0; // the input map the outputs were computed from
    std::unique_ptr<MapT> out;
    bool hit(size_t c) const { return valid && cycle == c; }
    bool hit(size_t c, const MapT &in) const { return hit(c) && in_id == in.id() && in_gen == in.generation(); }
};

// Device-resident replacement of HashMap<String, Ciphertext> (circuit.rs:517-520):
// wire name -> row of an HBM wire table owned by the engine context.
class EncWireMap {
  public:
    EncWireMap(helm_hip_ctx *ctx, int n) : ctx_(ctx), n_(n) {}
    ~EncWireMap();
    EncWireMap(const EncWireMap &) = delete;
    EncWireMap &operator=(const EncWireMap &) = delete;
    bool contains_key(const std::string &k) const { return index_.count(k) != 0; }
    size_t len() const { return index_.size(); }
    std::vector<std::string> keys() const;
    int row(const std::string &k) const; // throws Panic if absent
    std::vector<uint32_t> get(const std::string &k) const;        // download one ciphertext
    void insert(const std::string &k, const uint32_t *lwe);       // upload (adds the key if new)
    std::unique_ptr<EncWireMap> clone() const;                    // fresh owned map, device copy
    helm_hip_wires *table() const { return wires_; }
    helm_hip_ctx *ctx() const { return ctx_; }
    // build with a fixed key set (rows in iteration order of `names`)
    void reserve_keys(const std::vector<std::string> &names);
    int scratch(int64_t rows); // makes room for `rows` unnamed rows behind the named ones; their first row
    // identity of this map object and a counter of its host-side modifications: what the evaluators' same-cycle
    // memo (gates.rs:55-59) compares instead of the ciphertext rows themselves
    uint64_t id() const { return id_; }
    uint64_t generation() const { return gen_; }
    void touch() { gen_++; } // a caller that wrote rows through table() says so

  private:
    void grow(int64_t rows);
    const uint64_t id_ = next_map_id();
    uint64_t gen_ = 0;
    helm_hip_ctx *ctx_;
    int n_;
    helm_hip_wires *wires_ = nullptr;
    int64_t cap_ = 0;
    std::unordered_map<std::string, int> index_;
};

// reference src/circuit.rs:35-58
template <typename MapT> class EvalCircuit {
  public:
    virtual ~EvalCircuit() = default;
    virtual std::unique_ptr<MapT> encrypt_inputs(const std::set<std::string> &wire_set,
                                                 const std::map<std::string, PtxtType> &input_wire_map) = 0;
    virtual std::unique_ptr<MapT> evaluate_encrypted(const MapT &enc_wire_map, size_t current_cycle,
                                                     const std::string &ptxt_type) = 0;
    virtual std::unique_ptr<MapT> init_ready() = 0;
    virtual void evaluate_ready(const MapT &enc_wire_map, MapT &valid_outputs) = 0;
    virtual std::map<std::string, PtxtType> decrypt_outputs(const MapT &enc_wire_map, bool verbose) = 0;
};

// reference src/circuit.rs:69-73, 383-391, 449-577.  client_key: the CPU client;
// server_key: an engine context with both keys loaded.
class GateCircuit : public EvalCircuit<EncWireMap> {
  public:
    GateCircuit(helm_client_key *client_key, helm_hip_ctx *server_key, Circuit circuit);
    ~GateCircuit() override;
    std::unique_ptr<EncWireMap> encrypt_inputs(const std::set<std::string> &wire_set,
                                               const std::map<std::string, PtxtType> &input_wire_map) override;
    std::unique_ptr<EncWireMap> evaluate_encrypted(const EncWireMap &enc_wire_map, size_t current_cycle,
                                                   const std::string &ptxt_type) override;
    std::unique_ptr<EncWireMap> init_ready() override;
    void evaluate_ready(const EncWireMap &enc_wire_map, EncWireMap &valid_outputs) override;
    std::map<std::string, PtxtType> decrypt_outputs(const EncWireMap &enc_wire_map, bool verbose) override;

    const Circuit &circuit() const { return circuit_; }
    int64_t pbs_per_cycle() const { return pbs_count_; }
    std::string log() { std::string s; s.swap(log_); return s; } // progress lines (circuit.rs:542)
    // Multi-GPU (one process per GPU; keys, circuit and input ciphertexts the same on every rank): from the next
    // evaluate_encrypted on, the launches are packed for the communicator's world size and every launch of more than
    // `replicate_below` bootstraps is split over its ranks, the output ciphertexts all-gathered with ncclAllGather inside
    // the engine (helm_hip_program_run_sharded_comm, include/helm_comm.h) - the level of circuit.rs:531 is the sharded
    // unit.  Every rank ends with the wire map of a one-GPU evaluation.  comm = nullptr: back to one GPU.
    void shard_over(helm_comm *comm, int64_t replicate_below);
    // the exchange of a launch overlapped with the launches that do not need its outputs (run_sharded_comm, overlap = 1)
    void set_exchange_overlap(bool on) { overlap_ = on; }

  private:
    helm_client_key *client_key_;
    helm_hip_ctx *server_key_;
    Circuit circuit_;
    int n_;
    // program cache: rebuilt when the wire-name -> row mapping changes
    helm_hip_program *prog_ = nullptr;
    std::vector<std::string> prog_keys_;
    std::vector<int> prog_rows_;
    int64_t pbs_count_ = 0;
    int64_t prog_launches_ = 0;
    std::vector<int64_t> level_end_; // program level at which each circuit level ends (unpacked schedule)
    int64_t n_scratch_ = 0;          // scratch rows the program uses behind the named rows (copies for flip-flop chains)
    helm_comm *comm_ = nullptr;      // shard_over(): the engine's RCCL communicator, or null
    int comm_world_ = 1;
    int64_t replicate_below_ = 256;
    bool overlap_ = false;
    bool packed_ = false; // the program's launches are packed rounds, not the circuit's levels
    std::string log_;
    // Same cycle AND the very input map (unmodified): the cached wire map is returned without a launch.  The
    // reference's boolean path has its cache probe commented out (gates.rs:247-252), so the memo must not be
    // observable there: requiring unchanged inputs makes it return exactly what a re-evaluation would.
    CycleMemo<EncWireMap> memo_;
    int64_t memo_hits_ = 0;

  public:
    int64_t memo_hits() const { return memo_hits_; }
};

// ---------------------------------------------------------------------------------------
// LUT mode and arithmetic mode (include/helm_shortint.h)
// ---------------------------------------------------------------------------------------

// Device-resident replacement of HashMap<String, CtxtShortInt> / HashMap<String, FheType>
// (circuit.rs:1046-1049, 1312-1315): wire name -> first of `blocks` consecutive rows of a
// big-LWE table (1 row per shortint wire, 4..64 per FheUint8..128 wire).
class SiEncWireMap {
  public:
    SiEncWireMap(helm_si_ctx *ctx, int blocks);
    ~SiEncWireMap();
    SiEncWireMap(const SiEncWireMap &) = delete;
    SiEncWireMap &operator=(const SiEncWireMap &) = delete;
    bool contains_key(const std::string &k) const { return index_.count(k) != 0; }
    size_t len() const { return index_.size(); }
    int blocks() const { return blocks_; }
    int row_words() const { return dim_ + 1; }
    std::vector<std::string> keys() const;
    int row(const std::string &k) const;                    // first row; throws Panic if absent
    std::vector<uint64_t> get(const std::string &k) const;  // download `blocks` ciphertexts
    void insert(const std::string &k, const uint64_t *lwe); // upload (adds the key if new)
    std::unique_ptr<SiEncWireMap> clone(int64_t scratch_rows) const;
    void reserve_keys(const std::vector<std::string> &names, int64_t scratch_rows);
    int scratch(int64_t rows); // first row of a scratch region behind the named rows
    helm_si_wires *table() const { return wires_; }
    uint64_t id() const { return id_; }
    uint64_t generation() const { return gen_; }
    void touch() { gen_++; }

  private:
    void grow(int64_t rows);
    const uint64_t id_ = next_map_id();
    uint64_t gen_ = 0;
    helm_si_ctx *ctx_;
    int blocks_, dim_ = 0;
    helm_si_wires *wires_ = nullptr;
    int64_t cap_ = 0;
    std::unordered_map<std::string, int> index_;
};

// reference src/circuit.rs:75-79, 969-1120
class LutCircuit : public EvalCircuit<SiEncWireMap> {
  public:
    LutCircuit(helm_si_client_key *client_key, helm_si_ctx *server_key, Circuit circuit);
    std::unique_ptr<SiEncWireMap> encrypt_inputs(const std::set<std::string> &wire_set,
                                                 const std::map<std::string, PtxtType> &input_wire_map) override;
    std::unique_ptr<SiEncWireMap> evaluate_encrypted(const SiEncWireMap &enc_wire_map, size_t current_cycle,
                                                     const std::string &ptxt_type) override;
    std::unique_ptr<SiEncWireMap> init_ready() override;
    void evaluate_ready(const SiEncWireMap &enc_wire_map, SiEncWireMap &valid_outputs) override;
    std::map<std::string, PtxtType> decrypt_outputs(const SiEncWireMap &enc_wire_map, bool verbose) override;
    int64_t pbs_per_cycle() const { return pbs_count_; }
    std::string log() { std::string s; s.swap(log_); return s; }
    // Gate::evaluate_encrypted_high_precision_lut (gates.rs:721-742): LUT gates whose index does not fit one block
    // (more inputs than log2(message_modulus * carry_modulus)) go through the WoP-PBS path once a key is set
    void set_wide_lut_key(helm_wop_ctx *wop, int bits_per_block);
    // The reference prints `PBS time: {} us` for every LUT gate (gates.rs:293-302); here that is one line per LUT gate
    // carrying its level's time, which costs one host synchronisation per level.  set_timing_lines(false) drops the
    // lines AND the per-level synchronisation: the host then prepares the next level while the GPU runs this one.
    void set_timing_lines(bool on) { timing_lines_ = on; }

  private:
    helm_si_client_key *client_key_;
    helm_si_ctx *server_key_;
    helm_wop_ctx *wop_ = nullptr;
    int wop_bits_per_block_ = 1;
    bool timing_lines_ = true;
    Circuit circuit_;
    helm_si_params P_{};
    int64_t pbs_count_ = 0;
    std::string log_;
    // As GateCircuit's: same cycle and the very input map.  (The reference's evaluate_encrypted_lut probes its cache
    // but never stores `cycle`, gates.rs:282-304, so there it only ever hits for cycle 0.)
    CycleMemo<SiEncWireMap> memo_;
    int64_t memo_hits_ = 0;

  public:
    int64_t memo_hits() const { return memo_hits_; }
};

// One integer operator of a level: rows a, b, out are the first rows of radix integers.
struct RadixOp {
    enum Kind { Copy, Add, Sub, Mul, Div, Shl, Shr, AddScalar, SubScalar, MulScalar, DivScalar, ShlScalar, ShrScalar } kind = Copy;
    int a = -1, b = -1, out = -1;
    unsigned __int128 scalar = 0;
    // Carry-save form: an integer kept as the SUM of two rows of blocks <= 3 (what a multiplication's term reduction
    // leaves before its final carry propagation).  a2 / b2 >= 0: that operand's second term (Add, Sub, and block shifts =
    // MulScalar by a power of four); out2 >= 0: leave the result in that form (Mul, block shifts) - no propagation.
    int a2 = -1, b2 = -1, out2 = -1;
};

// Level-batched FheUintN operators (add, sub, mul and their scalar forms, copy) over the two
// device primitives helm_si_lincomb / helm_si_apply_luts.
// Round merger: operator chains that share no wire run on one host thread each and meet here with every look-up round.
// When each running chain has handed in its next round, the rounds are merged into launches of at most `capacity`
// ciphertexts (helm_si_round_capacity: what the device bootstraps at once - a launch of that size takes one bootstrap's
// time whatever it holds) on ONE context and stream: the chain with the most rounds left is served first, a round that does
// not fit is continued in the next launch next to the other chains' NEXT rounds.  The reference's unit is the level
// (src/circuit.rs:1321: every operator of a level, then a join); this is the same evaluation with the join per look-up
// round instead of per level - identical ciphertexts, the device never waits for a launch slot.
class RoundMerger {
  public:
    RoundMerger(helm_si_ctx *ctx, int chains, int64_t capacity, bool strict = false);
    void set_remaining(int chain, int64_t rounds) { remaining_[(size_t)chain] = rounds; }
    // the chain's next round; returns once all of it is ENQUEUED (stream order does the rest)
    void submit(int chain, helm_si_wires *w, const std::vector<int32_t> &in, const std::vector<int32_t> &lut,
                const std::vector<int32_t> &out, const uint64_t *luts, int64_t n_luts);
    void finish(int chain); // no more rounds from this chain (also on error)
    std::mutex &device() { return mu_; } // every other device call of a chain holds it: one context, several threads
    int64_t launches() const { return launches_; }

  private:
    struct Sub {
        helm_si_wires *w = nullptr;
        const std::vector<int32_t> *in = nullptr, *lut = nullptr, *out = nullptr;
        const uint64_t *luts = nullptr;
        int64_t n_luts = 0;
        size_t taken = 0;
        bool present = false;
    };
    void issue_locked();
    helm_si_ctx *ctx_;
    int64_t capacity_;
    bool strict_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Sub> subs_;
    std::vector<int64_t> remaining_;
    std::vector<char> active_;
    std::string error_;
    int64_t launches_ = 0;
};

class RadixEngine {
  public:
    RadixEngine(helm_si_ctx *ctx, int blocks);
    void attach(RoundMerger *merger, int chain) { merger_ = merger, chain_ = chain; }
    int64_t scratch_rows(const std::vector<RadixOp> &ops) const;
    void run_level(helm_si_wires *w, const std::vector<RadixOp> &ops_in, int scratch);
    void propagate(helm_si_wires *w, const std::vector<int32_t> &bases, int scratch, int width,
                   const std::vector<int32_t> *carry_out_rows);
    int64_t pbs_count() const { return pbs_count_; }
    int64_t pbs_rounds() const { return pbs_rounds_; }

  private:
    void lincomb(helm_si_wires *w, const std::vector<int32_t> &in_idx, const std::vector<int64_t> &coef,
                 const std::vector<int64_t> &cadd, const std::vector<int32_t> &out, int terms);
    void apply(helm_si_wires *w, const std::vector<int32_t> &in, const std::vector<int32_t> &lut,
               const std::vector<int32_t> &out);
    struct Carries {
        std::vector<int32_t> row; // [integer * (need + 1) + item]
        bool bit = true;          // 0 / 4, else c form (4 c: carry <=> c >= 2)
    };
    Carries carries(helm_si_wires *w, const std::vector<int32_t> &states, int G, int n, int need, bool add_const, int &sp,
                    bool want_bit);
    static int prop_rows(int W); // scratch rows of propagate() per integer
    void shift_scalar(helm_si_wires *w, const std::vector<RadixOp> &ops);
    void shift_encrypted(helm_si_wires *w, const std::vector<RadixOp> &ops, int &scratch_pos);
    void divide(helm_si_wires *w, const std::vector<RadixOp> &ops, int &scratch_pos);
    helm_si_ctx *ctx_;
    int nb_;
    helm_si_params P_{};
    std::vector<uint64_t> luts_;
    int lut_msg_ = 0, lut_carry_ = 0, lut_mul_lo_ = 0, lut_mul_hi_ = 0, lut_mul2_lo_ = 0, lut_mul2_hi_ = 0;
    int lut_t_[4] = {}, lut_s_[4] = {}, lut_q_[3] = {}, lut_gc_[3] = {}, lut_q3_ = 0, lut_gc3_ = 0, lut_resolve_ = 0, lut_final_ = 0,
        lut_cout_ = 0; // carry propagation, see propagate()
    int lut_bit0_ = 0, lut_bit1_ = 0, lut_shl1_ = 0, lut_shr1_ = 0, lut_sel_ = 0;
    int64_t pbs_count_ = 0, pbs_rounds_ = 0;
    std::vector<int32_t> pend_in_, pend_lut_, pend_out_; // look-ups waiting for the level's next batch
    RoundMerger *merger_ = nullptr; // set: look-up rounds go through the merger, other device calls under its lock
    int chain_ = 0;
    std::unique_lock<std::mutex> device_lock() { return merger_ ? std::unique_lock<std::mutex>(merger_->device()) : std::unique_lock<std::mutex>(); }
};

// reference src/circuit.rs:81-85, 1112-1500
class ArithCircuit : public EvalCircuit<SiEncWireMap> {
  public:
    ArithCircuit(helm_si_client_key *client_key, helm_si_ctx *server_key, Circuit circuit);
    ~ArithCircuit() override;
    std::unique_ptr<SiEncWireMap> encrypt_inputs(const std::set<std::string> &wire_set,
                                                 const std::map<std::string, PtxtType> &input_wire_map) override;
    std::unique_ptr<SiEncWireMap> evaluate_encrypted(const SiEncWireMap &enc_wire_map, size_t current_cycle,
                                                     const std::string &ptxt_type) override;
    std::unique_ptr<SiEncWireMap> init_ready() override;
    void evaluate_ready(const SiEncWireMap &enc_wire_map, SiEncWireMap &valid_outputs) override;
    std::map<std::string, PtxtType> decrypt_outputs(const SiEncWireMap &enc_wire_map, bool verbose) override;
    int64_t pbs_per_cycle() const { return pbs_count_; }
    int64_t pbs_rounds_per_cycle() const { return pbs_rounds_; }
    std::string log() { std::string s; s.swap(log_); return s; }
    // Lanes (helm_si_ctx_fork): sub-circuits that share no wire are evaluated concurrently, one lane each, instead of
    // meeting at every level boundary (circuit.rs:1321 joins the whole level).  Same ciphertexts, fewer rounds in a row.
    // Default: with two or more independent sub-circuits they run as chains on the server key's own context, their look-up
    // rounds merged (RoundMerger).  add_lane() installs the caller's lanes instead; clear_lanes() switches concurrency
    // off altogether (level by level, as the reference).
    // (every schedule switch also drops the same-cycle memo: a cycle evaluated under another schedule is not "the same call")
    void add_lane(helm_si_ctx *lane) { lanes_.push_back(lane); reset_memo(); }
    void clear_lanes() { lanes_.clear(); auto_lanes_ = false; reset_memo(); }
    void set_lazy_carries(bool on) { lazy_carries_ = on; reset_memo(); }
    // The same-cycle memo is keyed on the cycle ALONE, as the reference's (gates.rs:307-312): evaluate_encrypted with a
    // cycle that was already evaluated returns that cycle's gate outputs WHATEVER the inputs.  reset_memo() forgets the
    // remembered cycle (and releases the device copy of its wire map); set_memo(false) switches the memo off.
    void reset_memo() { memo_.valid = false; memo_.out.reset(); }
    void set_memo(bool on) { memo_on_ = on; if (!on) reset_memo(); }
    // Merged rounds: launches of at most this many ciphertexts (0 = helm_si_round_capacity(), what the device
    // bootstraps at once).  Tests force 1 to cut every round into single-ciphertext launches.
    void set_round_capacity(int64_t capacity) { round_capacity_ = capacity; reset_memo(); }

  private:
    void encrypt_value(SiEncWireMap &m, const std::string &wire, unsigned __int128 value);
    helm_si_client_key *client_key_;
    helm_si_ctx *server_key_;
    std::vector<helm_si_ctx *> lanes_;
    bool auto_lanes_ = true;
    bool lazy_carries_ = true; // carry-save products feeding additions / subtractions (set_lazy_carries)
    bool memo_on_ = true;
    int64_t round_capacity_ = 0;
    Circuit circuit_;
    helm_si_params P_{};
    std::string global_ptxt_type_;
    int64_t pbs_count_ = 0, pbs_rounds_ = 0;
    std::string log_;
    // The reference's arithmetic gates return their cached output whenever the cycle repeats, WHATEVER the inputs
    // (gates.rs:307-312; tests/gates_test.rs:196-223 calls again with other operands and expects the first result;
    // tests/circuit_test.rs:314-474 passes cycles 1..4 "to defeat the cache"): keyed on the cycle alone.
    CycleMemo<SiEncWireMap> memo_;
    int64_t memo_hits_ = 0;

  public:
    int64_t memo_hits() const { return memo_hits_; }
};

} // namespace helm
