// preprocessor.cpp — raw synthesis output -> the netlist dialect HELM's parser reads.
//
// The reference keeps this step in its benchmark submodule (`cargo run --bin preprocessor --
// --input raw.v --output processed.v [--arithmetic]`, reference README.md:116-120,133-137,
// RUNNING.md throughout); the submodule is not vendored (.gitmodules:1-4), so this is a
// restatement of its CONTRACT, written against what the parser on the other side accepts
// (src/verilog_parser.rs:31-276), not of its source:
//
//   structural mode   Yosys `write_verilog` netlists - comments and (* attributes *), declarations
//                     with several names per range, multi-line cell instances with named ports
//                     (.A(x), .Y(y)), `\`-escaped identifiers, parameterised `$lut` cells, `assign`
//                     aliases / constants / bitwise expressions, concatenations and part selects -
//                     become one gate per line, `kw NAME(in0, in1, out);`, output last, one name per
//                     ranged declaration, constants as cone / czero gates.
//   --arithmetic      behavioural Verilog made of `assign` statements over + - * / << >> becomes
//                     three-address add / sub / mult / div / shl / shr / copy gates on whole words;
//                     an all-digit operand is a plaintext scalar and always the SECOND operand's
//                     role (the evaluator computes ciphertext OP scalar whatever the position,
//                     src/circuit.rs:1328-1387), so `c - x`, `c / x`, `c << x` first build c as a word.
#include "helm_host.hpp"

#include <algorithm>
#include <cctype>
#include <functional>
#include <sstream>

namespace helm {
namespace {

struct Tok {
    enum Kind { Ident, Number, Punct, End } kind = End;
    std::string text;
};

std::string strip_comments(const std::string &s)
{
    std::string o;
    o.reserve(s.size());
    for (size_t i = 0; i < s.size();) {
        if (s.compare(i, 2, "//") == 0) {
            while (i < s.size() && s[i] != '\n') i++;
        } else if (s.compare(i, 2, "/*") == 0) {
            size_t e = s.find("*/", i + 2);
            i = e == std::string::npos ? s.size() : e + 2;
            o += ' ';
        } else if (s.compare(i, 2, "(*") == 0 && i + 2 < s.size() && s[i + 2] != ')') {
            size_t e = s.find("*)", i + 2); // (* src = "..." *) attribute
            i = e == std::string::npos ? s.size() : e + 2;
            o += ' ';
        } else
            o += s[i++];
    }
    return o;
}

std::vector<Tok> lex(const std::string &s)
{
    std::vector<Tok> t;
    for (size_t i = 0; i < s.size();) {
        const unsigned char c = (unsigned char)s[i];
        if (std::isspace(c)) {
            i++;
        } else if (c == '\\') { // escaped identifier: up to the next white space
            size_t e = i + 1;
            while (e < s.size() && !std::isspace((unsigned char)s[e])) e++;
            t.push_back({Tok::Ident, s.substr(i + 1, e - i - 1)});
            i = e;
        } else if (std::isalpha(c) || c == '_' || c == '$') {
            size_t e = i;
            while (e < s.size() && (std::isalnum((unsigned char)s[e]) || s[e] == '_' || s[e] == '$')) e++;
            t.push_back({Tok::Ident, s.substr(i, e - i)});
            i = e;
        } else if (std::isdigit(c) || c == '\'') {
            size_t e = i;
            while (e < s.size() && (std::isdigit((unsigned char)s[e]) || s[e] == '_')) e++;
            if (e < s.size() && s[e] == '\'') {
                e++;
                if (e < s.size() && (s[e] == 's' || s[e] == 'S')) e++;
                if (e < s.size()) e++; // base letter
                while (e < s.size() && (std::isalnum((unsigned char)s[e]) || s[e] == '_' || s[e] == '?')) e++;
            }
            t.push_back({Tok::Number, s.substr(i, e - i)});
            i = e;
        } else if ((c == '<' || c == '>') && i + 1 < s.size() && s[i + 1] == (char)c) {
            size_t e = i + 2;
            if (e < s.size() && s[e] == (char)c) e++; // <<< >>>
            t.push_back({Tok::Punct, std::string(2, (char)c)});
            i = e;
        } else if (c == '~' && i + 1 < s.size() && s[i + 1] == '^') {
            t.push_back({Tok::Punct, "~^"});
            i += 2;
        } else if (c == '^' && i + 1 < s.size() && s[i + 1] == '~') {
            t.push_back({Tok::Punct, "~^"});
            i += 2;
        } else {
            t.push_back({Tok::Punct, std::string(1, (char)c)});
            i++;
        }
    }
    t.push_back({Tok::End, ""});
    return t;
}

// value and width of a Verilog literal: 8'h96, 1'b0, 32'd3, 12
struct Literal {
    unsigned __int128 value = 0;
    int width = -1; // -1: unsized
};
Literal parse_literal(const std::string &s)
{
    Literal L;
    const size_t q = s.find('\'');
    std::string digits = s;
    int base = 10;
    if (q != std::string::npos) {
        if (q > 0) {
            std::string w = s.substr(0, q);
            w.erase(std::remove(w.begin(), w.end(), '_'), w.end());
            L.width = std::stoi(w);
        }
        size_t b = q + 1;
        if (b < s.size() && (s[b] == 's' || s[b] == 'S')) b++;
        const char bc = b < s.size() ? (char)std::tolower((unsigned char)s[b]) : 'd';
        base = bc == 'h' ? 16 : bc == 'b' ? 2 : bc == 'o' ? 8 : 10;
        digits = s.substr(b + 1);
    }
    for (char ch : digits) {
        if (ch == '_') continue;
        int d;
        if (std::isdigit((unsigned char)ch)) d = ch - '0';
        else if (std::isxdigit((unsigned char)ch)) d = std::tolower((unsigned char)ch) - 'a' + 10;
        else if (ch == 'x' || ch == 'X' || ch == 'z' || ch == 'Z' || ch == '?') d = 0; // don't-care reads as 0
        else throw Panic("preprocessor: bad literal '" + s + "'");
        if (d >= base) throw Panic("preprocessor: bad literal '" + s + "'");
        L.value = L.value * (unsigned)base + (unsigned)d;
    }
    return L;
}

std::string u128_text(unsigned __int128 v)
{
    if (v == 0) return "0";
    std::string s;
    while (v) {
        s.insert(s.begin(), (char)('0' + (int)(v % 10)));
        v /= 10;
    }
    return s;
}

std::string lower(std::string s)
{
    for (auto &c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}

class Parser {
  public:
    Parser(const std::string &text, bool arithmetic) : toks_(lex(strip_comments(text))), arith_(arithmetic) {}

    std::string run()
    {
        bool any = false;
        while (peek().kind != Tok::End) {
            if (peek().text == "module") {
                module();
                any = true;
            } else
                pos_++; // `timescale, stray tokens between modules
        }
        if (!any) throw Panic("preprocessor: no module found");
        return out_.str();
    }

  private:
    const Tok &peek(size_t k = 0) const { return toks_[std::min(pos_ + k, toks_.size() - 1)]; }
    Tok next() { return toks_[std::min(pos_++, toks_.size() - 1)]; }
    bool accept(const std::string &p)
    {
        if (peek().kind != Tok::End && peek().text == p && peek().kind != Tok::Number) {
            pos_++;
            return true;
        }
        return false;
    }
    void expect(const std::string &p)
    {
        if (!accept(p)) throw Panic("preprocessor: expected '" + p + "' near '" + peek().text + "'");
    }

    // ---- names ------------------------------------------------------------------------------
    // the dialect is split on ',' and ' ' and looks for '(' ')' ';' (verilog_parser.rs:172-175): every
    // other character of an escaped or generated name is mapped to '_', collisions get a suffix
    std::string clean(const std::string &raw)
    {
        auto it = clean_.find(raw);
        if (it != clean_.end()) return it->second;
        std::string c;
        for (char ch : raw) c += (std::isalnum((unsigned char)ch) || ch == '_' || ch == '[' || ch == ']') ? ch : '_';
        if (c.empty() || std::isdigit((unsigned char)c[0])) c = "n_" + c; // an all-digit name would read as a scalar
        std::string u = c;
        for (int k = 2; used_.count(u); k++) u = c + "_" + std::to_string(k);
        used_.insert(u);
        clean_[raw] = u;
        return u;
    }
    std::string fresh(const char *stem)
    {
        for (;;) {
            std::string n = std::string(stem) + std::to_string(tmp_++);
            if (!used_.count(n) && !clean_.count(n)) {
                used_.insert(n);
                return n;
            }
        }
    }
    void gate(const std::string &kw, const std::vector<std::string> &operands)
    {
        out_ << "  " << kw << " " << fresh("pp_g") << "(";
        for (size_t i = 0; i < operands.size(); i++) out_ << (i ? ", " : "") << operands[i];
        out_ << ");\n";
    }
    std::string const_wire(bool one)
    {
        std::string &w = one ? const1_ : const0_;
        if (w.empty()) {
            w = fresh(one ? "pp_const_one" : "pp_const_zero");
            gate(one ? "cone" : "czero", {w});
        }
        return w;
    }

    // ---- module --------------------------------------------------------------------------------
    void module()
    {
        expect("module");
        const std::string name = clean(next().text);
        std::vector<std::string> ports;
        std::ostringstream decl;
        if (accept("#")) skip_parens();
        if (accept("(")) {
            std::string dir;
            std::pair<int, int> range{-1, -1};
            while (!accept(")")) {
                if (accept(",")) continue;
                const std::string t = peek().text;
                if (t == "input" || t == "output" || t == "inout") {
                    dir = next().text;
                    range = {-1, -1};
                    while (peek().text == "wire" || peek().text == "reg" || peek().text == "signed" || peek().text == "logic") pos_++;
                    if (peek().text == "[") range = parse_range();
                    continue;
                }
                const std::string raw = next().text;
                ports.push_back(clean(raw));
                if (!dir.empty()) declare(dir, range, {raw}, decl);
            }
        }
        expect(";");
        out_ << "module " << name << "(";
        for (size_t i = 0; i < ports.size(); i++) out_ << (i ? ", " : "") << ports[i];
        out_ << ");\n" << decl.str();
        while (peek().kind != Tok::End && peek().text != "endmodule") item();
        expect("endmodule");
        out_ << "endmodule\n";
    }

    void skip_parens()
    {
        expect("(");
        for (int depth = 1; depth > 0;) {
            const Tok t = next();
            if (t.kind == Tok::End) throw Panic("preprocessor: unbalanced parentheses");
            if (t.kind == Tok::Punct && t.text == "(") depth++;
            if (t.kind == Tok::Punct && t.text == ")") depth--;
        }
    }

    std::pair<int, int> parse_range()
    {
        expect("[");
        const int hi = (int)const_expr();
        expect(":");
        const int lo = (int)const_expr();
        expect("]");
        return {hi, lo};
    }
    long const_expr()
    { // literal [+|- literal]: ranges such as [WIDTH-1:0] are not supported, [8-1:0] is
        long v = (long)parse_literal(next().text).value;
        while (peek().text == "+" || peek().text == "-") {
            const bool plus = next().text == "+";
            const long w = (long)parse_literal(next().text).value;
            v = plus ? v + w : v - w;
        }
        return v;
    }

    void declare(const std::string &dir, std::pair<int, int> range, const std::vector<std::string> &raws, std::ostringstream &os)
    {
        for (auto &raw : raws) {
            const std::string n = clean(raw);
            if (range.first >= 0) width_[n] = range;
            if (dir == "wire" || dir == "reg") {
                os << "  wire " << n << ";\n"; // parsed and ignored by the reference (verilog_parser.rs:217-221)
                continue;
            }
            // one name per ranged declaration: the parser reads tokens[2] only (verilog_parser.rs:178-188)
            if (range.first >= 0) os << "  " << dir << " [" << range.first << ":" << range.second << "] " << n << ";\n";
            else os << "  " << dir << " " << n << ";\n";
        }
    }

    void item()
    {
        const std::string t = peek().text;
        if (t == "input" || t == "output" || t == "inout" || t == "wire" || t == "reg") {
            const std::string dir = next().text == "inout" ? "input" : t;
            while (peek().text == "wire" || peek().text == "reg" || peek().text == "signed") pos_++;
            std::pair<int, int> range{-1, -1};
            if (peek().text == "[") range = parse_range();
            std::vector<std::string> names;
            while (!accept(";")) {
                if (accept(",")) continue;
                if (accept("=")) { // wire x = expr;
                    const std::string lhs = names.back();
                    declare(dir, range, {lhs}, out_);
                    names.pop_back();
                    assign_to(arith_ ? Bits{clean(lhs)} : bits_of_name(clean(lhs)));
                    continue;
                }
                names.push_back(next().text);
            }
            declare(dir, range, names, out_);
        } else if (t == "assign") {
            next();
            do {
                const std::vector<std::string> lhs = arith_ ? std::vector<std::string>{clean(next().text)} : operand();
                expect("=");
                assign_to(lhs);
            } while (accept(","));
            expect(";");
        } else if (t == "parameter" || t == "localparam" || t == "specify" || t == "always" || t == "initial" ||
                   t == "function" || t == "generate")
            throw Panic("preprocessor: '" + t + "' is not a netlist construct (run synthesis first)");
        else
            cell();
    }

    // ---- nets -> single-bit names, most significant first --------------------------------------
    using Bits = std::vector<std::string>;
    std::vector<std::string> bits_of_name(const std::string &n)
    {
        auto it = width_.find(n);
        if (it == width_.end()) return {n};
        std::vector<std::string> b;
        const int hi = it->second.first, lo = it->second.second;
        for (int i = hi; hi >= lo ? i >= lo : i <= lo; i += hi >= lo ? -1 : 1) b.push_back(n + "[" + std::to_string(i) + "]");
        return b;
    }
    std::vector<std::string> operand()
    {
        if (accept("{")) {
            std::vector<std::string> b;
            if (peek().kind == Tok::Number && peek(1).text == "{") { // replication {3{x}}
                const int rep = (int)parse_literal(next().text).value;
                expect("{");
                std::vector<std::string> inner;
                do {
                    auto p = e_cond();
                    inner.insert(inner.end(), p.begin(), p.end());
                } while (accept(","));
                expect("}");
                expect("}");
                for (int r = 0; r < rep; r++) b.insert(b.end(), inner.begin(), inner.end());
                return b;
            }
            do {
                auto p = e_cond(); // concatenation items may be expressions
                b.insert(b.end(), p.begin(), p.end());
            } while (accept(","));
            expect("}");
            return b;
        }
        if (peek().kind == Tok::Number) {
            const Literal L = parse_literal(next().text);
            const int w = L.width > 0 ? L.width : 32;
            std::vector<std::string> b;
            for (int i = w - 1; i >= 0; i--) b.push_back(((L.value >> i) & 1) ? "$1" : "$0");
            return b;
        }
        if (peek().kind != Tok::Ident) throw Panic("preprocessor: expected a net near '" + peek().text + "'");
        const std::string n = clean(next().text);
        if (accept("[")) {
            const int a = (int)const_expr();
            if (accept(":")) {
                const int b = (int)const_expr();
                expect("]");
                std::vector<std::string> v;
                for (int i = a; a >= b ? i >= b : i <= b; i += a >= b ? -1 : 1) v.push_back(n + "[" + std::to_string(i) + "]");
                return v;
            }
            expect("]");
            return {n + "[" + std::to_string(a) + "]"};
        }
        return bits_of_name(n);
    }
    // a single-bit net usable as a gate input: constants become cone / czero wires
    std::string in_bit(const std::string &b) { return b == "$0" ? const_wire(false) : b == "$1" ? const_wire(true) : b; }

    // ---- bitwise expressions of assign statements (structural mode) --------------------------
    Bits match(Bits a, size_t w)
    { // zero-extend / truncate on the most significant side
        while (a.size() < w) a.insert(a.begin(), "$0");
        while (a.size() > w) a.erase(a.begin());
        return a;
    }
    Bits binary(const char *kw, Bits a, Bits b)
    {
        const size_t w = std::max(a.size(), b.size());
        a = match(a, w);
        b = match(b, w);
        Bits r;
        for (size_t i = 0; i < w; i++) {
            r.push_back(fresh("pp_w"));
            gate(kw, {in_bit(a[i]), in_bit(b[i]), r.back()});
        }
        return r;
    }
    Bits e_primary()
    {
        if (accept("(")) {
            Bits b = e_cond();
            expect(")");
            return b;
        }
        if (accept("~") || accept("!")) {
            Bits a = e_primary(), r;
            for (auto &x : a) {
                if (x == "$0" || x == "$1") {
                    r.push_back(x == "$0" ? "$1" : "$0");
                    continue;
                }
                r.push_back(fresh("pp_w"));
                gate("not", {x, r.back()});
            }
            return r;
        }
        return operand();
    }
    Bits e_and()
    {
        Bits a = e_primary();
        while (peek().text == "&" && peek().kind == Tok::Punct) {
            next();
            a = binary("and", a, e_primary());
        }
        return a;
    }
    Bits e_xor()
    {
        Bits a = e_and();
        while (peek().kind == Tok::Punct && (peek().text == "^" || peek().text == "~^")) {
            const bool xnor = next().text == "~^";
            a = binary(xnor ? "xnor" : "xor", a, e_and());
        }
        return a;
    }
    Bits e_or()
    {
        Bits a = e_xor();
        while (peek().text == "|" && peek().kind == Tok::Punct) {
            next();
            a = binary("or", a, e_xor());
        }
        return a;
    }
    Bits e_cond()
    {
        Bits c = e_or();
        if (!accept("?")) return c;
        Bits t = e_cond();
        expect(":");
        Bits e = e_cond();
        if (c.size() != 1) throw Panic("preprocessor: the condition of ?: must be one bit wide");
        const size_t w = std::max(t.size(), e.size());
        t = match(t, w);
        e = match(e, w);
        Bits r;
        for (size_t i = 0; i < w; i++) {
            r.push_back(fresh("pp_w"));
            gate("mux", {in_bit(t[i]), in_bit(e[i]), in_bit(c[0]), r.back()}); // out = sel ? in0 : in1 (gates.rs:189-192)
        }
        return r;
    }

    void assign_to(const Bits &lhs)
    {
        if (arith_) {
            arith_assign(lhs[0]);
            return;
        }
        Bits rhs = match(e_cond(), lhs.size());
        for (size_t i = 0; i < lhs.size(); i++) {
            if (rhs[i] == "$0" || rhs[i] == "$1") gate(rhs[i] == "$1" ? "cone" : "czero", {lhs[i]});
            else gate("buf", {rhs[i], lhs[i]});
        }
    }

    // ---- cell instances -----------------------------------------------------------------------
    // $_AND_, AND2_X1, NAND2, sky130_fd_sc_hd__and2_1, \$lut, DFFPOSX1 -> the dialect's keyword ("" = unknown)
    static std::string cell_kind(const std::string &raw)
    {
        const std::string t = lower(raw);
        if (t == "$lut" || t == "lut") return "lut";
        static const char *known[] = {"xnor", "nand", "xor", "nor", "and", "or", "not", "inv", "buf", "mux", "dff"};
        std::string run;
        for (size_t i = 0; i <= t.size(); i++) {
            if (i < t.size() && std::isalpha((unsigned char)t[i])) {
                run += t[i];
                continue;
            }
            for (const char *k : known)
                if (run == k) return run == "inv" ? "not" : run;
            run.clear();
        }
        if (t.find("dff") != std::string::npos) return "dff";
        return "";
    }

    void cell()
    {
        const std::string type_raw = next().text;
        std::map<std::string, Literal> params;
        if (accept("#")) {
            expect("(");
            while (!accept(")")) {
                if (accept(",")) continue;
                if (accept(".")) {
                    const std::string pn = next().text;
                    expect("(");
                    params[lower(pn)] = parse_literal(next().text);
                    expect(")");
                } else
                    params["#" + std::to_string(params.size())] = parse_literal(next().text);
            }
        }
        if (peek().kind != Tok::Ident) throw Panic("preprocessor: cannot read the statement starting at '" + type_raw + "'");
        const std::string inst = clean(next().text);
        const std::string kind = cell_kind(type_raw);
        if (kind.empty()) throw Panic("preprocessor: unknown cell type '" + type_raw + "' (instance " + inst + ")");
        expect("(");
        std::vector<std::pair<std::string, Bits>> named;
        std::vector<Bits> positional;
        while (!accept(")")) {
            if (accept(",")) continue;
            if (accept(".")) {
                const std::string port = lower(next().text);
                expect("(");
                Bits b;
                if (peek().text != ")") b = operand();
                expect(")");
                named.push_back({port, b});
            } else
                positional.push_back(operand());
        }
        expect(";");

        Bits ins;
        std::string outw;
        auto one = [&](const Bits &b, const std::string &what) -> std::string {
            if (b.size() != 1) throw Panic("preprocessor: port " + what + " of " + inst + " is not one bit wide");
            return b[0];
        };
        if (!named.empty()) {
            static const std::set<std::string> outs = {"y", "q", "z", "zn", "o", "out", "x"};
            static const std::set<std::string> clocks = {"c", "clk", "ck", "clock", "cp", "gclk"};
            std::vector<std::pair<std::string, Bits>> in_ports;
            for (auto &kv : named) {
                if (outs.count(kv.first) && outw.empty()) outw = one(kv.second, kv.first);
                else if (kind == "dff" && kv.first != "d") {
                    if (!clocks.count(kv.first)) throw Panic("preprocessor: DFF port '" + kv.first + "' (set / reset / enable) has no counterpart in the dialect");
                } else
                    in_ports.push_back(kv);
            }
            if (kind == "mux") { // Yosys $_MUX_: Y = S ? B : A  ->  mux(in0 = B, in1 = A, sel = S)
                std::map<std::string, Bits> m(in_ports.begin(), in_ports.end());
                auto pick = [&](std::initializer_list<const char *> names) -> Bits {
                    for (auto n : names)
                        if (m.count(n)) return m[n];
                    throw Panic("preprocessor: MUX " + inst + " needs ports A, B, S");
                };
                ins = {one(pick({"b", "i1", "d1", "a2"}), "B"), one(pick({"a", "i0", "d0", "a1"}), "A"), one(pick({"s", "sel", "s0"}), "S")};
            } else if (kind == "lut") {
                for (auto &kv : in_ports) ins.insert(ins.end(), kv.second.begin(), kv.second.end()); // .A({msb, ..., lsb})
            } else {
                std::sort(in_ports.begin(), in_ports.end(), [](auto &a, auto &b) { return a.first < b.first; }); // A, B, C ... / A1, A2 / I0, I1
                for (auto &kv : in_ports) ins.push_back(one(kv.second, kv.first));
            }
        } else {
            // positional connections: Verilog gate primitives, output first
            if (positional.size() < 2) throw Panic("preprocessor: instance " + inst + " has too few connections");
            outw = one(positional[0], "0");
            for (size_t i = 1; i < positional.size(); i++) ins.push_back(one(positional[i], std::to_string(i)));
        }
        if (outw.empty()) throw Panic("preprocessor: instance " + inst + " has no output port (Y / Q / Z / O)");
        if (outw == "$0" || outw == "$1") throw Panic("preprocessor: instance " + inst + " drives a constant");

        out_ << "  ";
        if (kind == "lut") {
            auto it = params.find("lut");
            if (it == params.end()) it = params.find("init");
            if (it == params.end()) throw Panic("preprocessor: LUT " + inst + " has no LUT / INIT parameter");
            // $lut: Y = LUT[A], A[0] the last element of the concatenation - the dialect's index convention
            // (first input = most significant bit, gates.rs:159-167; table bit i = (const >> i) & 1)
            out_ << "lut " << inst << "(0x";
            static const char *hx = "0123456789abcdef";
            std::string h;
            for (unsigned __int128 v = it->second.value; v; v >>= 4) h.insert(h.begin(), hx[(int)(v & 15)]);
            out_ << (h.empty() ? "0" : h);
            for (auto &b : ins) out_ << ", " << in_bit_inline(b);
            out_ << ", " << outw << ");\n";
            flush_pending();
            return;
        }
        const size_t want = kind == "not" || kind == "buf" || kind == "dff" ? 1 : kind == "mux" ? 3 : 2;
        if (ins.size() < want) throw Panic("preprocessor: instance " + inst + " (" + kind + ") has too few inputs");
        if (ins.size() > want && kind != "mux") {
            // wider library gates (AND3, NOR4): the dialect's gates are two-input - build a chain
            std::string acc = in_bit_inline(ins[0]);
            const std::string base = kind == "nand" ? "and" : kind == "nor" ? "or" : kind == "xnor" ? "xor" : kind;
            std::ostringstream chain;
            for (size_t i = 1; i < ins.size(); i++) {
                const bool last = i + 1 == ins.size();
                const std::string o = last ? outw : fresh("pp_w");
                chain << (i > 1 ? "  " : "") << (last ? kind : base) << " " << (last ? inst : fresh("pp_g")) << "(" << acc << ", "
                      << in_bit_inline(ins[i]) << ", " << o << ");\n";
                acc = o;
            }
            out_ << chain.str();
            flush_pending();
            return;
        }
        out_ << kind << " " << inst << "(";
        for (auto &b : ins) out_ << in_bit_inline(b) << ", ";
        out_ << outw << ");\n";
        flush_pending();
    }
    // constants used by the gate being written: their cone / czero gates are emitted right after it
    std::string in_bit_inline(const std::string &b)
    {
        if (b != "$0" && b != "$1") return b;
        std::string &w = b == "$1" ? const1_ : const0_;
        if (w.empty()) {
            w = fresh(b == "$1" ? "pp_const_one" : "pp_const_zero");
            pending_ += std::string("  ") + (b == "$1" ? "cone " : "czero ") + fresh("pp_g") + "(" + w + ");\n";
        }
        return w;
    }
    void flush_pending()
    {
        out_ << pending_;
        pending_.clear();
    }

    // ---- arithmetic mode: three-address code over whole words ----------------------------------
    struct Val {
        bool lit = false;
        unsigned __int128 v = 0;
        std::string wire;
        std::string text() const { return lit ? u128_text(v) : wire; }
    };
    Val a_primary()
    {
        if (accept("(")) {
            Val v = a_shift();
            expect(")");
            return v;
        }
        if (peek().kind == Tok::Number) {
            Val v;
            v.lit = true;
            v.v = parse_literal(next().text).value;
            return v;
        }
        if (peek().kind != Tok::Ident) throw Panic("preprocessor: expected an operand near '" + peek().text + "'");
        Val v;
        v.wire = clean(next().text);
        return v;
    }
    Val as_word(const Val &v, const Val &some_wire)
    { // a literal as a ciphertext word: (w - w) + c
        const std::string z = fresh("pp_t"), k = fresh("pp_t");
        stmt_ops_.push_back({"sub", some_wire.wire, some_wire.wire, z});
        stmt_ops_.push_back({"add", z, u128_text(v.v), k});
        Val r;
        r.wire = k;
        return r;
    }
    Val a_emit(const std::string &op, Val a, Val b)
    {
        if (a.lit && b.lit) { // constant folding, modulo nothing: the evaluator truncates scalars to the word width
            Val r;
            r.lit = true;
            if (op == "add") r.v = a.v + b.v;
            else if (op == "sub") r.v = a.v - b.v;
            else if (op == "mult") r.v = a.v * b.v;
            else if (op == "div") r.v = b.v ? a.v / b.v : ~(unsigned __int128)0;
            else if (op == "shl") r.v = b.v < 128 ? a.v << (int)b.v : 0;
            else r.v = b.v < 128 ? a.v >> (int)b.v : 0;
            return r;
        }
        if (a.lit) {
            if (op == "add" || op == "mult") std::swap(a, b);
            else a = as_word(a, b);
        }
        Val r;
        r.wire = fresh("pp_t");
        stmt_ops_.push_back({op, a.text(), b.text(), r.wire});
        return r;
    }
    Val a_mul()
    {
        Val a = a_primary();
        while (peek().kind == Tok::Punct && (peek().text == "*" || peek().text == "/")) {
            const bool mul = next().text == "*";
            a = a_emit(mul ? "mult" : "div", a, a_primary());
        }
        return a;
    }
    Val a_add()
    {
        Val a = a_mul();
        while (peek().kind == Tok::Punct && (peek().text == "+" || peek().text == "-")) {
            const bool add = next().text == "+";
            a = a_emit(add ? "add" : "sub", a, a_mul());
        }
        return a;
    }
    Val a_shift()
    {
        Val a = a_add();
        while (peek().kind == Tok::Punct && (peek().text == "<<" || peek().text == ">>")) {
            const bool left = next().text == "<<";
            a = a_emit(left ? "shl" : "shr", a, a_add());
        }
        return a;
    }
    void arith_assign(const std::string &lhs)
    {
        stmt_ops_.clear();
        const Val v = a_shift();
        if (v.lit) throw Panic("preprocessor: '" + lhs + "' is assigned a constant: the arithmetic dialect has no constant gate");
        // the statement's last operation writes the left-hand side itself; a bare `assign y = x;` is a copy
        const bool direct = !stmt_ops_.empty() && stmt_ops_.back()[3] == v.wire;
        if (direct) stmt_ops_.back()[3] = lhs;
        for (auto &o : stmt_ops_) gate(o[0], {o[1], o[2], o[3]});
        if (!direct) gate("copy", {v.wire, lhs});
    }

    std::vector<Tok> toks_;
    size_t pos_ = 0;
    bool arith_;
    std::ostringstream out_;
    std::map<std::string, std::string> clean_;
    std::set<std::string> used_;
    std::map<std::string, std::pair<int, int>> width_;
    std::vector<std::vector<std::string>> stmt_ops_; // arithmetic mode: (op, a, b, out) of the statement being lowered
    std::string const0_, const1_, pending_;
    int tmp_ = 0;
};

} // namespace

std::string preprocess(const std::string &text, bool arithmetic) { return Parser(text, arithmetic).run(); }

} // namespace helm
