// bristol.hpp -- Bristol netlist front end: analyzer + "assembler" text writer.
//
// Same interface and file formats as the reference's analyze_bristol (src/analyze.h:86-87,
// src/analyze.cpp:56) and assemble_bristol (src/assemble.h:43-44, src/assemble.cpp:46);
// re-implemented with O(G) bookkeeping (the reference's fan-in/out pass is O(V*G) and its
// register lookup O(G) per operand).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace bce {

// src/analyze.h:40-54
struct Variable {
    std::string in_fname;
    bool new_flag = false;
    unsigned n_tot = 0;        // circuit nodes (wires)
    unsigned n_inputs = 0;
    unsigned n_in1_bits = 0;
    unsigned n_in2_bits = 0;
    unsigned n_out1_bits = 0;   // ALL output bits (the buses concatenated, first output first)
    // Bristol Fashion allows any number of input / output values: every bus width, in header order
    // (the reference hard-codes two inputs and one output, src/analyze.cpp:129-158)
    std::vector<unsigned> in_bits, out_bits;
    std::vector<unsigned> high_water, low_water, life, fan_in, fan_out;
};

// src/analyze.h:58-72
struct Function {
    std::string in_fname;
    uint64_t n_tot = 0;        // function calls (gates)
    // "XOR" / "AND" / "NOT" / "EQW" (wire copy) / " EQ" (constant: in_list[i][0] is the literal 0 or 1, not a wire);
    // a Bristol Fashion MAND line is expanded into its single ANDs
    std::vector<std::string> call_list;
    std::vector<std::vector<unsigned>> in_list, out_list;
    unsigned n_and = 0, n_or = 0, n_xor = 0, n_not = 0, n_eq = 0, n_eqw = 0;
    std::vector<std::string> names;
};

struct Analysis {
    Variable variables;
    Function functions;
};

// throws std::runtime_error on unreadable / malformed input (the reference exits the process)
Analysis analyze_bristol(const std::string& in_fname, bool gen_fan_flag, bool new_flag, bool quiet = false);

// writes the assembler program; returns the path written.
// out_path empty: "<in_fname up to the first '.'>_FHE.out" (max_depth 0) as in the reference.
std::string assemble_bristol(const Analysis& analysis, unsigned max_depth, bool debug_flag,
                             const std::string& out_path = std::string(), bool quiet = false);

}  // namespace bce
