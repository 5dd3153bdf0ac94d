// bristol.cpp -- see bristol.hpp.  Formats: SURVEY.md App. A (assembler text) and App. B
// (old / new "Bristol Fashion" netlists).
#include "bristol.hpp"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

namespace bce {

namespace {
bool next_line(std::ifstream& f, std::string& line) { return static_cast<bool>(std::getline(f, line)); }
}

Analysis analyze_bristol(const std::string& in_fname, bool gen_fan_flag, bool new_flag, bool quiet) {
    std::ifstream f(in_fname);
    if (!f) throw std::runtime_error("analyze_bristol: error opening file " + in_fname);
    std::string line;
    Analysis A;
    Variable& v = A.variables;
    Function& fn = A.functions;
    v.in_fname = fn.in_fname = in_fname;
    v.new_flag = new_flag;

    unsigned n_func = 0, n_var = 0;
    if (!next_line(f, line) || !(std::istringstream(line) >> n_func >> n_var))
        throw std::runtime_error("analyze_bristol: bad first header line in " + in_fname);
    v.n_inputs = 2;
    if (new_flag) {
        // "<n_inputs> <w1> [<w2> ...]" / "<n_outputs> <w1> ..." / blank  (src/analyze.cpp:129-158 reads two input
        // widths and one output width; Bristol Fashion files may carry any number of either)
        if (!next_line(f, line)) throw std::runtime_error("analyze_bristol: truncated header");
        {
            std::istringstream s(line);
            unsigned w = 0;
            if (!(s >> v.n_inputs) || v.n_inputs < 1) throw std::runtime_error("analyze_bristol: bad input header line in " + in_fname);
            for (unsigned k = 0; k < v.n_inputs; ++k) {
                if (!(s >> w)) throw std::runtime_error("analyze_bristol: input header names " + std::to_string(v.n_inputs) + " values but lists fewer widths");
                v.in_bits.push_back(w);
            }
        }
        if (!next_line(f, line)) throw std::runtime_error("analyze_bristol: truncated header");
        {
            // outputs are the last sum(w) wires, first output first
            std::istringstream so(line);
            unsigned n_out = 0, w = 0;
            if (!(so >> n_out) || n_out < 1) throw std::runtime_error("analyze_bristol: bad output header line in " + in_fname);
            for (unsigned k = 0; k < n_out; ++k) {
                if (!(so >> w)) throw std::runtime_error("analyze_bristol: output header lists fewer widths than values");
                v.out_bits.push_back(w);
            }
        }
        next_line(f, line);
    } else {
        // "<n_in1> <n_in2> <n_out>" / blank  (src/analyze.cpp:160-179)
        unsigned a = 0, b = 0, o = 0;
        if (!next_line(f, line) || !(std::istringstream(line) >> a >> b >> o))
            throw std::runtime_error("analyze_bristol: bad second header line in " + in_fname);
        v.in_bits = {a, b};
        v.out_bits = {o};
        next_line(f, line);
    }
    v.n_in1_bits = v.in_bits.empty() ? 0 : v.in_bits[0];
    v.n_in2_bits = v.in_bits.size() > 1 ? v.in_bits[1] : 0;
    v.n_out1_bits = 0;
    for (unsigned w : v.out_bits) v.n_out1_bits += w;
    {
        uint64_t tot_in = 0;
        for (unsigned w : v.in_bits) tot_in += w;
        if (tot_in > n_var || v.n_out1_bits > n_var)
            throw std::runtime_error("analyze_bristol: header widths exceed the wire count");
    }
    v.n_tot = n_var;
    fn.n_tot = n_func;
    fn.names = {"XOR", "AND", "NOT", " EQ", "EQW"};
    v.high_water.assign(n_var, 0);
    v.low_water.assign(n_var, 0);
    v.life.assign(n_var, 0);
    v.fan_in.assign(n_var, 0);
    v.fan_out.assign(n_var, 0);
    fn.call_list.reserve(n_func);
    fn.in_list.reserve(n_func);
    fn.out_list.reserve(n_func);

    for (unsigned ix = 0; ix < n_func; ++ix) {
        do {
            if (!next_line(f, line)) throw std::runtime_error("analyze_bristol: file ends before gate " + std::to_string(ix));
        } while (line.find_first_not_of(" \t\r\n") == std::string::npos);
        std::istringstream s(line);
        unsigned nin = 0, nout = 0;
        s >> nin >> nout;
        std::vector<unsigned> il(nin), ol(nout);
        for (auto& w : il) s >> w;
        for (auto& w : ol) s >> w;
        std::string op;
        s >> op;
        if (!s) throw std::runtime_error("analyze_bristol: bad gate line " + std::to_string(ix));
        for (auto& ch : op) ch = (char)std::toupper((unsigned char)ch);
        const bool is_eq = op == "EQ";   // "1 1 <0|1> <out> EQ": the input field is a literal, not a wire
        if (!is_eq) for (unsigned w : il) if (w >= n_var) throw std::runtime_error("analyze_bristol: wire index out of range");
        for (unsigned w : ol) if (w >= n_var) throw std::runtime_error("analyze_bristol: wire index out of range");
        // water marks, with the reference's "0 means unset" convention (src/analyze.cpp:288-301)
        auto touch = [&](unsigned w) {
            if (v.low_water[w] == 0) v.low_water[w] = ix;
            v.high_water[w] = ix;
        };
        auto add = [&](const char* name, std::vector<unsigned> ins, std::vector<unsigned> outs, bool wires) {
            if (wires) for (unsigned w : ins) { touch(w); if (gen_fan_flag) ++v.fan_out[w]; }
            for (unsigned w : outs) { touch(w); if (gen_fan_flag) ++v.fan_in[w]; }
            fn.call_list.emplace_back(name);
            fn.in_list.push_back(std::move(ins));
            fn.out_list.push_back(std::move(outs));
        };
        if (op == "XOR" || op == "AND") {
            if (nin != 2 || nout != 1) throw std::runtime_error("analyze_bristol: " + op + " needs 2 inputs and 1 output (line " + std::to_string(ix) + ")");
            ++(op == "XOR" ? fn.n_xor : fn.n_and);
            add(op.c_str(), il, ol, true);
        } else if (op == "INV" || op == "NOT") {
            if (nin != 1 || nout != 1) throw std::runtime_error("analyze_bristol: INV needs 1 input and 1 output (line " + std::to_string(ix) + ")");
            ++fn.n_not;
            add("NOT", il, ol, true);
        } else if (op == "EQW") {
            if (nin != 1 || nout != 1) throw std::runtime_error("analyze_bristol: EQW needs 1 input and 1 output (line " + std::to_string(ix) + ")");
            ++fn.n_eqw;
            add("EQW", il, ol, true);
        } else if (is_eq) {
            // constant assignment (the reference gives up here: "Cannot parse EQ!! yet failing", src/analyze.cpp:273-277)
            if (nin != 1 || nout != 1 || il[0] > 1) throw std::runtime_error("analyze_bristol: EQ takes the literal 0 or 1 and one output wire (line " + std::to_string(ix) + ")");
            ++fn.n_eq;
            add(" EQ", il, ol, false);
        } else if (op == "MAND") {
            // multiple AND: out[k] = in[k] AND in[m + k]; expanded into m two-input ANDs
            if (nout == 0 || nin != 2 * nout) throw std::runtime_error("analyze_bristol: MAND needs 2m inputs and m outputs (line " + std::to_string(ix) + ")");
            for (unsigned k = 0; k < nout; ++k) {
                ++fn.n_and;
                add("AND", {il[k], il[nout + k]}, {ol[k]}, true);
            }
        } else {
            throw std::runtime_error("analyze_bristol: bad parse of function on line " + std::to_string(ix));
        }
    }
    fn.n_tot = fn.call_list.size();
    for (unsigned w = 0; w < n_var; ++w) v.life[w] = v.high_water[w] - v.low_water[w];
    if (!gen_fan_flag) { v.fan_in.clear(); v.fan_out.clear(); }
    if (!quiet) {
        std::cout << "Analysis Report for input file " << in_fname << "\n"
                  << "Total number of nodes: " << n_var << "\n"
                  << "number bits input 1 = " << v.n_in1_bits << "\n"
                  << "number bits input 2 = " << v.n_in2_bits << "\n"
                  << "number bits output 1 = " << v.n_out1_bits << "\n"
                  << "Total number of function calls " << fn.n_tot << "\n"
                  << " number of and " << fn.n_and << "\n number of xor " << fn.n_xor << "\n number of inv " << fn.n_not
                  << "\n number of eq " << fn.n_eq << "\n number of weqw " << fn.n_eqw << std::endl;
        if (gen_fan_flag && n_var) {
            std::cout << "max fan in (should be 1) = " << *std::max_element(v.fan_in.begin(), v.fan_in.end()) << "\n"
                      << "max fan out = " << *std::max_element(v.fan_out.begin(), v.fan_out.end()) << std::endl;
        }
        if (n_var) std::cout << "max variable life = " << *std::max_element(v.life.begin(), v.life.end()) << std::endl;
    }
    return A;
}

std::string assemble_bristol(const Analysis& analysis, unsigned max_depth, bool debug_flag, const std::string& out_path,
                             bool quiet) {
    const Variable& v = analysis.variables;
    const Function& f = analysis.functions;
    std::string fname = out_path;
    if (fname.empty()) {
        fname = v.in_fname.substr(0, v.in_fname.find("."));
        fname += (max_depth == 0) ? std::string("_FHE.out") : "_" + std::to_string(max_depth) + ".out";
    }
    if (max_depth == 0) max_depth = 10000;
    FILE* fid = std::fopen(fname.c_str(), "w");
    if (!fid) throw std::runtime_error("assemble_bristol: error opening output file " + fname);
    if (!quiet) std::cout << "Assembler: opening output file " << fname << " for output" << std::endl;

    std::fprintf(fid, "# Max depth %d\n", max_depth);
    std::fprintf(fid, "# number input1 bits %d\n", v.n_in1_bits);
    std::fprintf(fid, "# number input2 bits %d\n", v.n_in2_bits);
    std::fprintf(fid, "# number output1 bits %d\n", v.n_out1_bits);
    for (size_t k = 2; k < v.in_bits.size(); ++k) std::fprintf(fid, "# number input%d bits %d\n", (int)k + 1, v.in_bits[k]);
    if (v.out_bits.size() > 1) {
        std::fprintf(fid, "# output buses");
        for (unsigned w : v.out_bits) std::fprintf(fid, " %d", w);
        std::fprintf(fid, "\n");
    }

    // registers are never recycled (src/assemble.cpp:212-225): register k = k-th defined node
    std::vector<int> node_reg(v.n_tot, -1);
    unsigned reg = 0;
    auto load = [&](unsigned bus, unsigned bits, unsigned first_node) {
        for (unsigned ix = 0; ix < bits; ++ix, ++reg) {
            std::fprintf(fid, "R%d = LOAD(In%d,%d)\n", reg, bus, ix);
            node_reg[first_node + ix] = (int)reg;
            if (debug_flag) std::fprintf(fid, "# Assigned node %d to R%d\n", first_node + ix, reg);
        }
    };
    {   // input bus k occupies the next in_bits[k] nodes (In3, In4, ... for Bristol Fashion files with more values)
        unsigned first = 0;
        for (size_t k = 0; k < v.in_bits.size(); ++k) { load((unsigned)k + 1, v.in_bits[k], first); first += v.in_bits[k]; }
    }

    const unsigned first_out = v.n_tot - v.n_out1_bits;  // outputs are the last nodes (src/assemble.cpp:187-193)
    std::vector<int> out_reg(v.n_out1_bits, -1);
    for (size_t line_ix = 0; line_ix < f.call_list.size(); ++line_ix) {
        if (reg >= v.n_tot) {
            std::fclose(fid);
            throw std::runtime_error("assemble_bristol: ran out of register storage");
        }
        const unsigned out_node = f.out_list[line_ix].at(0);
        if (f.call_list[line_ix] == "EQW") {
            // wire copy: the output node is another name of the input's register (the reference's assembler writes a
            // parse-error comment here, src/assemble.cpp:370-373, and leaves the node undefined)
            const int r = node_reg[f.in_list[line_ix].at(0)];
            if (r < 0) { std::fclose(fid); throw std::runtime_error("assemble_bristol: EQW of an undefined node"); }
            node_reg[out_node] = r;
            if (debug_flag) std::fprintf(fid, "# Assigned node %d to R%d\n", out_node, r);
            if (out_node >= first_out) out_reg[out_node - first_out] = r;
            continue;
        }
        const unsigned out_r = reg++;
        node_reg[out_node] = (int)out_r;
        if (debug_flag) std::fprintf(fid, "# Assigned node %d to R%d\n", out_node, out_r);
        std::vector<int> in_r;
        const std::string& name = f.call_list[line_ix];
        if (name == " EQ") {
            // constant register -- an extension of the text format (the reference's assembler has no such line)
            std::fprintf(fid, "R%d = CONST(%d)\n", out_r, (int)f.in_list[line_ix].at(0));
            if (out_node >= first_out) {
                std::fprintf(fid, "# R%d is a terminal output register for out%ld\n", out_r, (long)(out_node - first_out));
                out_reg[out_node - first_out] = (int)out_r;
            }
            continue;
        }
        for (unsigned w : f.in_list[line_ix]) {
            if (node_reg[w] < 0) {
                std::fclose(fid);
                throw std::runtime_error("assemble_bristol: input register not found for node " + std::to_string(w));
            }
            in_r.push_back(node_reg[w]);
        }
        unsigned depth = 0;
        if (name == "XOR" && in_r.size() == 2) {
            depth = 1;
            std::fprintf(fid, "R%d = %s(R%d, R%d)  !depth = %d\n", out_r, name.c_str(), in_r[0], in_r[1], depth);
        } else if (name == "AND" && in_r.size() == 2) {
            depth = 1;
            std::fprintf(fid, "R%d = %s(R%d, R%d) !depth = %d\n", out_r, name.c_str(), in_r[0], in_r[1], depth);
        } else if (name == "NOT" && in_r.size() == 1) {
            std::fprintf(fid, "R%d = %s(R%d) !depth = %d\n", out_r, name.c_str(), in_r[0], depth);
        } else {
            if (!quiet) std::cout << "parse error on line :" << line_ix << std::endl;
            std::fprintf(fid, "#parse error on line %d\n", (int)line_ix);
        }
        if (out_node >= first_out) {
            const unsigned o = out_node - first_out;
            std::fprintf(fid, "# R%d is a terminal output register for out%ld\n", out_r, (long)o);
            out_reg[o] = (int)out_r;
        }
    }
    for (unsigned o = 0; o < v.n_out1_bits; ++o) {
        int r = out_reg[o] >= 0 ? out_reg[o] : node_reg[first_out + o];  // an output may be an input node
        std::fprintf(fid, "Out%d = STORE(R%d) ! depth = %d\n", o, r, 0);
    }
    std::fprintf(fid, "# Assembler statistics\n");
    std::fprintf(fid, "# max depth supported: %d\n", max_depth);
    std::fprintf(fid, "# max depth required: %d\n", 0);
    std::fprintf(fid, "# max tower jump: %d\n", 0);
    std::fprintf(fid, "# %d registers used\n", reg);
    std::fprintf(fid, "# %d BOOT operations required\n", 0);
    std::fclose(fid);
    return fname;
}

}  // namespace bce
