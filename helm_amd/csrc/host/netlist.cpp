// netlist.cpp — Gate, verilog_parser, input-map helpers and the plaintext Circuit.
// C++ mirror of reference src/gates.rs, src/verilog_parser.rs, src/lib.rs:90-194 and
// src/circuit.rs:104-381; see helm_host.hpp.
#include "helm_host.hpp"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

namespace helm {

static std::string u128_to_string(unsigned __int128 v)
{
    if (v == 0) return "0";
    std::string s;
    while (v) {
        s.push_back((char)('0' + (int)(v % 10)));
        v /= 10;
    }
    std::reverse(s.begin(), s.end());
    return s;
}

std::string PtxtType::to_string() const
{
    switch (kind) {
    case None: return "None";
    case Bool: return value ? "true" : "false";
    default: return u128_to_string(value);
    }
}

const char *gate_type_name(GateType t)
{
    static const char *names[] = {"And", "Dff", "Lut", "Mux", "Nand", "Nor", "Not", "Or", "Xnor", "Xor", "Buf",
                                  "ConstOne", "ConstZero", "Mult", "Add", "Sub", "Div", "Shl", "Shr", "Copy"};
    return names[(int)t];
}

static bool expect_bool(const PtxtType &v)
{
    if (v.kind != PtxtType::Bool) throw Panic("Expected PtxtType::Bool variant");
    return v.as_bool();
}

// reference src/gates.rs:151-239
PtxtType Gate::evaluate(const std::vector<PtxtType> &in)
{
    auto all = [&] { bool r = true; for (auto &v : in) r = expect_bool(v) && r; return r; };
    auto any = [&] { bool r = false; for (auto &v : in) r = expect_bool(v) || r; return r; };
    auto ones = [&] { size_t c = 0; for (auto &v : in) c += expect_bool(v) ? 1 : 0; return c; };
    switch (gate_type_) {
    case GateType::Dff: output_ = in.at(0); break;
    case GateType::And: output_ = PtxtType::boolean(all()); break;
    case GateType::Lut: {
        // first input is the most significant index bit (gates.rs:159-167)
        size_t shift_amt = 0;
        const size_t end = in.size() - 1;
        for (size_t idx = 0; idx < in.size(); idx++)
            if (expect_bool(in[idx])) shift_amt += (size_t)1 << (end - idx);
        if (!lut_const_) throw Panic("Lut const not provided");
        output_ = PtxtType::boolean((lut_const_->at(shift_amt) & 1) > 0);
        break;
    }
    case GateType::Mult: case GateType::Div: case GateType::Add: case GateType::Sub:
    case GateType::Shl: case GateType::Shr: case GateType::Copy:
        throw Panic("internal error: entered unreachable code"); // unreachable!() in the reference
    case GateType::Mux: {
        const bool select = expect_bool(in.at(2)), in0 = expect_bool(in.at(0)), in1 = expect_bool(in.at(1));
        output_ = PtxtType::boolean((select && in0) || (!select && in1));
        break;
    }
    case GateType::Nand: output_ = PtxtType::boolean(!all()); break;
    case GateType::Nor: output_ = PtxtType::boolean(!any()); break;
    case GateType::Not: output_ = PtxtType::boolean(!expect_bool(in.at(0))); break;
    case GateType::Or: output_ = PtxtType::boolean(any()); break;
    case GateType::Xnor: output_ = PtxtType::boolean(ones() % 2 != 1); break;
    case GateType::Xor: output_ = PtxtType::boolean(ones() % 2 == 1); break;
    case GateType::Buf: output_ = in.at(0); break;
    case GateType::ConstOne: output_ = PtxtType::boolean(true); break;
    case GateType::ConstZero: output_ = PtxtType::boolean(false); break;
    }
    return output_;
}

std::string Gate::debug() const
{
    std::ostringstream os;
    os << gate_name_ << ": \"" << output_wire_ << "\"(" << output_.to_string() << ") = " << gate_type_name(gate_type_)
       << "([";
    for (size_t i = 0; i < input_wires_.size(); i++) os << (i ? ", " : "") << '"' << input_wires_[i] << '"';
    os << "]). Level " << level_;
    return os.str();
}

// ---------------------------------------------------------------------------------------
// string helpers with Rust semantics
// ---------------------------------------------------------------------------------------
static std::string trim(const std::string &s)
{
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}
static std::string trim_end_matches(std::string s, char c)
{
    while (!s.empty() && s.back() == c) s.pop_back();
    return s;
}
static std::string trim_matches(std::string s, char c)
{
    s = trim_end_matches(s, c);
    size_t a = 0;
    while (a < s.size() && s[a] == c) a++;
    return s.substr(a);
}
static bool starts_with(const std::string &s, const char *p) { return s.rfind(p, 0) == 0; }
static std::vector<std::string> split_any(const std::string &s, const char *seps, bool drop_empty)
{
    std::vector<std::string> out;
    std::string cur;
    for (char ch : s) {
        if (strchr(seps, ch)) {
            if (!drop_empty || !cur.empty()) out.push_back(cur);
            cur.clear();
        } else
            cur.push_back(ch);
    }
    if (!drop_empty || !cur.empty()) out.push_back(cur);
    return out;
}
// str::parse::<uN>(): decimal digits (optional leading '+'), no whitespace, range checked
static bool parse_unsigned(const std::string &s, int bits, unsigned __int128 &out)
{
    size_t i = 0;
    if (!s.empty() && s[0] == '+') i = 1;
    if (i >= s.size()) return false;
    unsigned __int128 v = 0;
    const unsigned __int128 max = bits == 128 ? ~(unsigned __int128)0 : (((unsigned __int128)1 << bits) - 1);
    for (; i < s.size(); i++) {
        if (s[i] < '0' || s[i] > '9') return false;
        const unsigned d = (unsigned)(s[i] - '0');
        if (v > (max - d) / 10) return false;
        v = v * 10 + d;
    }
    out = v;
    return true;
}
static bool is_u32_literal(const std::string &s)
{
    unsigned __int128 v;
    return parse_unsigned(s, 32, v);
}

// ---------------------------------------------------------------------------------------
// verilog_parser
// ---------------------------------------------------------------------------------------
namespace verilog_parser {

static std::string extract_const_val(const std::string &s) // verilog_parser.rs:10-18
{
    const size_t a = s.find('(');
    if (a == std::string::npos) throw Panic("Opening parenthesis not found");
    const size_t b = s.find(')', a + 1);
    if (b == std::string::npos) throw Panic("Closing parenthesis not found");
    return s.substr(a + 1, b - a - 1);
}

static std::vector<uint64_t> usize_to_bitvec(uint64_t value, size_t lut_size) // :20-29
{
    std::vector<uint64_t> bits;
    for (size_t i = 0; i < lut_size; i++) bits.push_back(i < 64 ? (value >> i) & 1 : 0);
    return bits;
}

Gate parse_gate(const std::vector<std::string> &tokens)
{
    static const std::map<std::string, GateType> kw = {
        {"and", GateType::And},   {"lut", GateType::Lut},   {"dff", GateType::Dff},     {"mux", GateType::Mux},
        {"nand", GateType::Nand}, {"nor", GateType::Nor},   {"not", GateType::Not},     {"or", GateType::Or},
        {"xnor", GateType::Xnor}, {"xor", GateType::Xor},   {"buf", GateType::Buf},     {"czero", GateType::ConstZero},
        {"cone", GateType::ConstOne}, {"add", GateType::Add}, {"mult", GateType::Mult}, {"div", GateType::Div},
        {"sub", GateType::Sub},   {"shl", GateType::Shl},   {"shr", GateType::Shr},     {"copy", GateType::Copy}};
    auto it = kw.find(tokens.at(0));
    if (it == kw.end()) throw Panic("Invalid gate type \"" + tokens[0] + "\"");
    const GateType gate_type = it->second;
    auto tok = [&](size_t i) -> const std::string & {
        if (i >= tokens.size()) throw Panic("index out of bounds: malformed gate line");
        return tokens[i];
    };
    std::vector<std::string> name_and_inputs;
    for (auto &s : split_any(tok(1), "(,", false))
        if (!trim(s).empty()) name_and_inputs.push_back(s);
    auto nai = [&](size_t i) -> const std::string & {
        if (i >= name_and_inputs.size()) throw Panic("index out of bounds: malformed gate line");
        return name_and_inputs[i];
    };
    auto out_tok = [&](const std::string &t) { return trim_end_matches(trim_end_matches(t, ';'), ')'); };
    const std::string gate_name = nai(0);
    std::vector<std::string> input_wires;
    std::string output_wire;
    switch (gate_type) {
    case GateType::Not: case GateType::Dff: case GateType::Buf:
        input_wires = {trim(nai(1))};
        output_wire = out_tok(tok(2));
        break;
    case GateType::Mux: case GateType::Lut:
        input_wires = {nai(1)};
        for (size_t i = 2; i + 1 < tokens.size(); i++) input_wires.push_back(trim(trim_end_matches(tokens[i], ',')));
        output_wire = out_tok(tokens.back());
        break;
    case GateType::ConstOne: case GateType::ConstZero:
        output_wire = extract_const_val(tok(1));
        break;
    case GateType::Copy:
        input_wires = {nai(1)};
        output_wire = out_tok(tok(2));
        break;
    default:
        input_wires = {nai(1), trim(trim_end_matches(tok(2), ','))};
        output_wire = out_tok(tok(3));
        break;
    }
    std::optional<std::vector<uint64_t>> lut_const;
    if (gate_type == GateType::Lut) {
        const std::string c = input_wires.front();
        input_wires.erase(input_wires.begin());
        uint64_t v = 0;
        if (starts_with(c, "0x")) {
            std::string h = c;
            while (starts_with(h, "0x")) h = h.substr(2);
            if (h.empty() || h.size() > 16) throw Panic("Failed to parse hex");
            for (char ch : h) {
                int d = ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10
                        : ch >= 'A' && ch <= 'F' ? ch - 'A' + 10 : -1;
                if (d < 0) throw Panic("Failed to parse hex");
                v = (v << 4) | (uint64_t)d;
            }
        } else {
            unsigned __int128 w;
            if (!parse_unsigned(c, 64, w)) throw Panic("Failed to parse integer");
            v = (uint64_t)w;
        }
        lut_const = usize_to_bitvec(v, (size_t)1 << input_wires.size());
    }
    return Gate(gate_name, gate_type, input_wires, lut_const, output_wire, 0);
}

std::optional<std::pair<size_t, size_t>> parse_range(const std::string &range_str)
{
    std::string t = range_str;
    while (!t.empty() && (t.front() == '[' || t.front() == ']')) t.erase(t.begin());
    while (!t.empty() && (t.back() == '[' || t.back() == ']')) t.pop_back();
    auto parts = split_any(t, ":", false);
    unsigned __int128 a, b;
    if (parts.empty() || !parse_unsigned(parts[0], 64, a)) return std::nullopt;
    if (parts.size() < 2 || !parse_unsigned(parts[1], 64, b)) b = a;
    return std::make_pair((size_t)std::min(a, b), (size_t)std::max(a, b));
}

static Netlist parse_stream(std::istream &in, bool is_arith)
{
    Netlist nl;
    std::string raw;
    auto clean = [](const std::string &t) { return trim_end_matches(trim_matches(t, ','), ';'); };
    while (std::getline(in, raw)) {
        const std::string line = trim(raw);
        if (line.empty() || starts_with(line, "module") || starts_with(line, "endmodule") || starts_with(line, "//"))
            continue;
        const std::vector<std::string> tokens = split_any(line, ", ", true);
        if (tokens.empty()) continue;
        if (tokens[0] == "input" || tokens[0] == "output") {
            auto &dst = tokens[0] == "input" ? nl.inputs : nl.outputs;
            if (tokens.size() < 2) throw Panic("index out of bounds: malformed declaration");
            if (auto r = parse_range(tokens[1])) {
                if (tokens.size() < 3) throw Panic("index out of bounds: malformed declaration");
                const std::string name = clean(tokens[2]);
                if (is_arith)
                    for (size_t i = 2; i < tokens.size(); i++) dst.push_back(clean(tokens[i]));
                else
                    for (size_t i = r->first; i <= r->second; i++) dst.push_back(name + "[" + std::to_string(i) + "]");
            } else
                for (size_t i = 1; i < tokens.size(); i++) dst.push_back(clean(tokens[i]));
        } else if (tokens[0] == "wire") {
            // parsed and ignored (verilog_parser.rs:217-221)
        } else {
            Gate gate = parse_gate(tokens);
            const GateType t = gate.get_gate_type();
            if (t == GateType::Dff) {
                nl.inputs.push_back(gate.get_output_wire());
                nl.dff_outputs.push_back(gate.get_output_wire());
            } else if (t == GateType::Lut)
                nl.has_luts = true;
            else if (t == GateType::Add || t == GateType::Sub || t == GateType::Mult || t == GateType::Div ||
                     t == GateType::Shl || t == GateType::Shr || t == GateType::Copy)
                nl.has_arith = true;
            nl.wire_set.insert(gate.get_output_wire());
            nl.gates.emplace(gate.get_gate_name(), gate); // HashSet::insert: first one wins
        }
    }
    if (nl.has_arith && nl.gates.empty()) throw Panic("[!] Parser error, no arithmetic gates detected.");
    if (nl.gates.empty())
        throw Panic("[!] Parser error, no gates detected. Make sure to use the 'no-expr' flag in Yosys.");
    if (nl.has_arith && nl.has_luts) throw Panic("Can't mix LUTs with arithmetic operators!");
    return nl;
}

Netlist read_verilog_file(const std::string &file_name, bool is_arith)
{
    std::ifstream f(file_name);
    if (!f) throw Panic("Failed to open file: " + file_name);
    return parse_stream(f, is_arith);
}

Netlist read_verilog_text(const std::string &text, bool is_arith)
{
    std::istringstream f(text);
    return parse_stream(f, is_arith);
}

std::map<std::string, PtxtType> read_input_wires(const std::string &file_name, const std::string &ptxt_type)
{
    std::ifstream f(file_name);
    if (!f) throw Panic("Failed to open CSV file: " + file_name);
    std::map<std::string, PtxtType> input_map;
    std::string line;
    bool header = true;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue; // the csv crate skips empty lines
        if (header) { // csv::Reader default: the first record is a header
            header = false;
            continue;
        }
        const std::vector<std::string> record = split_any(line, ",", false);
        const std::string wire_name = trim(record[0]);
        if (record.size() == 2)
            input_map[wire_name] = parse_input_wire(trim(record[1]), ptxt_type);
        else if (record.size() == 3 && ptxt_type == "bool") {
            unsigned __int128 width;
            if (!parse_unsigned(trim(record[2]), 64, width)) throw Panic("invalid wire width '" + record[2] + "'");
            if (width > 1) {
                std::string bits = hex_to_bitstring(trim(record[1]));
                std::reverse(bits.begin(), bits.end());
                for (size_t idx = 0; idx < (size_t)width; idx++)
                    input_map[wire_name + "[" + std::to_string(idx) + "]"] =
                        PtxtType::boolean(idx < bits.size() && bits[idx] == '1'); // zero padded
            } else
                input_map[wire_name] = parse_input_wire(trim(record[1]), ptxt_type);
        } else
            throw Panic("The CSV should contain either two or three columns");
    }
    return input_map;
}

void write_output_wires(const std::optional<std::string> &file_name, const std::map<std::string, PtxtType> &m)
{
    if (!file_name) return;
    std::ofstream f(*file_name);
    if (!f) throw Panic("Failed to create CSV file: " + *file_name);
    for (auto &kv : m) {
        if (kv.second.kind == PtxtType::None) throw Panic("internal error: entered unreachable code");
        f << kv.first << ", " << kv.second.to_string() << "\n";
    }
    std::cout << "Decrypted outputs written to " << *file_name << std::endl;
}

} // namespace verilog_parser

// reference src/lib.rs:90-106
PtxtType parse_input_wire(const std::string &wire, const std::string &ptxt_type)
{
    if (ptxt_type == "bool") {
        const std::string t = trim(wire);
        return PtxtType::boolean(t == "1" || t == "true"); // "1" | parse::<bool>().unwrap_or(false)
    }
    static const std::map<std::string, std::pair<PtxtType::Kind, int>> kinds = {
        {"u8", {PtxtType::U8, 8}},   {"u16", {PtxtType::U16, 16}},   {"u32", {PtxtType::U32, 32}},
        {"u64", {PtxtType::U64, 64}}, {"u128", {PtxtType::U128, 128}}};
    auto it = kinds.find(ptxt_type);
    if (it == kinds.end()) throw Panic("internal error: entered unreachable code");
    unsigned __int128 v;
    if (!parse_unsigned(wire, it->second.second, v))
        throw Panic("called `Result::unwrap()` on an `Err` value: ParseIntError (\"" + wire + "\" as " + ptxt_type + ")");
    return PtxtType{it->second.first, v};
}

// reference src/lib.rs:181-194
std::string hex_to_bitstring(const std::string &hex)
{
    std::string bits;
    for (char ch : hex) {
        int d = ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10
                : ch >= 'A' && ch <= 'F' ? ch - 'A' + 10 : -1;
        if (d < 0) throw Panic("internal error: entered unreachable code");
        for (int b = 3; b >= 0; b--) bits.push_back(((d >> b) & 1) ? '1' : '0');
    }
    return bits;
}

// reference src/lib.rs:113-179
std::map<std::string, PtxtType> get_input_wire_map(const std::optional<std::string> &inputs_filename,
                                                   const std::vector<std::vector<std::string>> &wire_inputs,
                                                   const std::string &arithmetic_type)
{
    if (inputs_filename) {
        std::cout << "[✓] Input wires were provided." << std::endl;
        return verilog_parser::read_input_wires(*inputs_filename, arithmetic_type);
    }
    if (!wire_inputs.empty()) {
        std::cout << "[✓] Input wires were provided." << std::endl;
        std::map<std::string, PtxtType> m;
        for (auto &parts : wire_inputs) {
            if (parts.size() == 2)
                m[parts[0]] = parse_input_wire(parts[1], arithmetic_type);
            else if (parts.size() == 3 && arithmetic_type == "bool") {
                unsigned __int128 width;
                if (!parse_unsigned(trim(parts[2]), 64, width)) throw Panic("invalid wire width '" + parts[2] + "'");
                std::string bits = hex_to_bitstring(trim(parts[1]));
                std::reverse(bits.begin(), bits.end());
                for (size_t idx = 0; idx < (size_t)width; idx++)
                    m[parts[0] + "[" + std::to_string(idx) + "]"] = PtxtType::boolean(idx < bits.size() && bits[idx] == '1');
            } else
                throw Panic("-w input should contain either two or three values");
        }
        return m;
    }
    std::cout << "[!] No input wires specified, they will be initialized to false." << std::endl;
    return {{"dummy", parse_input_wire("0", arithmetic_type)}};
}

// ---------------------------------------------------------------------------------------
// Circuit
// ---------------------------------------------------------------------------------------
// reference src/circuit.rs:122-171
void Circuit::sort_circuit()
{
    if (gates_.empty()) throw Panic("assertion failed: !self.gates.is_empty()");
    if (!ordered_gates_.empty()) throw Panic("assertion failed: self.ordered_gates.is_empty()");
    std::set<std::string> wire_status(input_wires_.begin(), input_wires_.end());
    std::vector<Gate> dff_level, const_level;
    while (!gates_.empty()) {
        std::vector<Gate> level;
        std::set<std::string> next_wire_status;
        bool progressed = false;
        for (auto it = gates_.begin(); it != gates_.end();) {
            const Gate &gate = it->second;
            bool ready;
            if (gate.get_gate_type() == GateType::Dff) {
                next_wire_status.insert(gate.get_output_wire());
                dff_level.push_back(gate);
                ready = true;
            } else if (gate.get_gate_type() == GateType::ConstOne || gate.get_gate_type() == GateType::ConstZero) {
                next_wire_status.insert(gate.get_output_wire());
                const_level.push_back(gate);
                ready = true;
            } else {
                ready = true;
                for (auto &w : gate.get_input_wires())
                    if (!wire_status.count(w) && !is_u32_literal(w)) {
                        ready = false;
                        break;
                    }
                if (ready) {
                    next_wire_status.insert(gate.get_output_wire());
                    level.push_back(gate);
                }
            }
            if (ready) {
                it = gates_.erase(it);
                progressed = true;
            } else
                ++it;
        }
        // the reference spins forever on an unresolvable netlist; fail instead
        if (!progressed) throw Panic("sort_circuit: " + std::to_string(gates_.size()) +
                                     " gate(s) depend on undriven wires or form a combinational loop (first: " +
                                     gates_.begin()->second.get_gate_name() + ")");
        wire_status.insert(next_wire_status.begin(), next_wire_status.end());
        std::sort(level.begin(), level.end()); // by gate name
        ordered_gates_.insert(ordered_gates_.end(), level.begin(), level.end());
    }
    // DEVIATION (documented in DESIGN.md): the reference marks cone/czero gates ready but
    // never emits them (circuit.rs:142-147 vs :167), so a constant wire keeps its initial
    // value and any gate reading it panics in compute_levels.  Here constants are real
    // level-0 gates, emitted first.
    std::sort(const_level.begin(), const_level.end());
    ordered_gates_.insert(ordered_gates_.begin(), const_level.begin(), const_level.end());
    ordered_gates_.insert(ordered_gates_.end(), dff_level.begin(), dff_level.end());
    gates_.clear();
}

// reference src/circuit.rs:174-239
void Circuit::compute_levels()
{
    if (!gates_.empty()) throw Panic("assertion failed: self.gates.is_empty()");
    if (ordered_gates_.empty()) throw Panic("assertion failed: !self.ordered_gates.is_empty()");
    constexpr size_t DFF_KEY = (size_t)-1; // std::usize::MAX
    std::map<std::string, size_t> wire_levels;
    for (auto &w : input_wires_) wire_levels[w] = 0;
    for (auto &gate : ordered_gates_) {
        if (gate.get_gate_type() == GateType::Dff) {
            gate.set_level(DFF_KEY);
            level_map_[DFF_KEY].push_back(gate);
            continue;
        }
        size_t depth = 0;
        for (auto &input : gate.get_input_wires()) {
            size_t input_depth;
            auto it = wire_levels.find(input);
            if (it != wire_levels.end()) input_depth = it->second;
            else if (is_u32_literal(input)) input_depth = 0;
            else throw Panic("Input " + input + " not found in wire_levels");
            depth = std::max(depth, input_depth + 1);
        }
        gate.set_level(depth);
        level_map_[depth].push_back(gate);
        wire_levels[gate.get_output_wire()] = depth;
    }
    // move the DFFs to the last level
    const size_t total_keys = level_map_.size();
    auto it = level_map_.find(DFF_KEY);
    if (it != level_map_.end()) {
        std::vector<Gate> dffs = std::move(it->second);
        level_map_.erase(it);
        for (auto &g : dffs) g.set_level(total_keys);
        level_map_[total_keys] = std::move(dffs);
    }
    ordered_gates_.clear();
}

// reference src/circuit.rs:245-333
std::map<std::string, PtxtType> Circuit::initialize_wire_map(const std::set<std::string> &wire_set,
                                                             const std::map<std::string, PtxtType> &user_inputs,
                                                             const std::string &ptxt_type) const
{
    auto zero = [&]() -> PtxtType {
        if (ptxt_type == "bool") return PtxtType::boolean(false);
        if (ptxt_type == "u8") return {PtxtType::U8, 0};
        if (ptxt_type == "u16") return {PtxtType::U16, 0};
        if (ptxt_type == "u32") return {PtxtType::U32, 0};
        if (ptxt_type == "u64") return {PtxtType::U64, 0};
        if (ptxt_type == "u128") return {PtxtType::U128, 0};
        throw Panic("internal error: entered unreachable code");
    };
    std::map<std::string, PtxtType> wire_map;
    for (auto &k : wire_set) wire_map[k] = PtxtType::none();
    for (auto &input_wire : input_wires_) {
        if (user_inputs.empty()) wire_map[input_wire] = zero();
        else {
            auto it = user_inputs.find(input_wire);
            if (it == user_inputs.end()) throw Panic("\n Input wire \"" + input_wire + "\" not in input wires!");
            (void)zero(); // validates ptxt_type like the reference's match
            wire_map[input_wire] = it->second;
        }
    }
    for (auto &w : dff_outputs_) wire_map[w] = zero();
    return wire_map;
}

std::string Circuit::print_level_map() const
{
    std::ostringstream os;
    for (auto &kv : level_map_) {
        os << "Level " << kv.first << ":\n";
        for (auto &g : kv.second) os << "  " << g.debug() << "\n";
    }
    return os.str();
}

// reference src/circuit.rs:348-381
std::map<std::string, PtxtType> Circuit::evaluate(const std::map<std::string, PtxtType> &wire_map)
{
    if (!gates_.empty()) throw Panic("assertion failed: self.gates.is_empty()");
    if (!ordered_gates_.empty()) throw Panic("assertion failed: self.ordered_gates.is_empty()");
    std::map<std::string, PtxtType> eval_values = wire_map;
    for (auto &kv : level_map_) {
        // gates of one level are independent; snapshot semantics as under the RwLocks
        std::vector<std::pair<const std::string *, PtxtType>> results;
        for (auto &gate : kv.second) {
            std::vector<PtxtType> input_values;
            for (auto &input : gate.get_input_wires()) {
                auto it = eval_values.find(input);
                if (it == eval_values.end()) throw Panic("wire \"" + input + "\" not in the wire map");
                input_values.push_back(it->second);
            }
            results.emplace_back(&gate.get_output_wire(), gate.evaluate(input_values));
        }
        for (auto &r : results) {
            auto it = eval_values.find(*r.first);
            if (it == eval_values.end()) throw Panic("wire \"" + *r.first + "\" not in the wire map");
            it->second = r.second;
        }
    }
    return eval_values;
}

} // namespace helm
