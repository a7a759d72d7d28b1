// Jacobian connectivity (dRdWCon), colouring and assembly maps - host graph work, one-off per mesh.
#pragma once
#include "das_common.hpp"

namespace das {

enum StateKind { KIND_VEC = 0, KIND_SCL = 1, KIND_FACE = 2 };

struct StateDef {
    std::string name;
    StateKind kind;
    long long offset;  // DAIndex "state" ordering offset (reference DAIndex.C:188-258)
    long long size;
};

// stateResConInfo: per residual block, per level, bitmask over states
struct Stencil {
    std::vector<StateDef> states;
    std::vector<std::vector<unsigned>> levels;  // levels[resBlock][lv] = state bitmask
    long long n = 0;
};

Stencil make_stencil(int solver, int nC, int nF, const Options& opt, bool isPC);

struct JacCon {
    long long n = 0, nnz = 0;
    std::vector<long long> rowptr;  // rows = residuals
    std::vector<int> col;           // sorted state indices
    std::vector<int> anchor;        // per row: the cell it is anchored at (cell rows: the cell; face rows: owner)
    // transposed structure (rows = states j, cols = residual i), CSR
    std::vector<long long> t_rowptr;
    std::vector<int> t_col;
    // assembly map: for pattern entry e (row-major), rc_dest[e] = index into the transposed value
    // array, entries of a row sorted by rc_color (the colour of their column)
    std::vector<unsigned short> rc_color;
    std::vector<unsigned> rc_dest;
    void build(const Mesh& m, const Stencil& st);
    void build_transpose_and_maps(const std::vector<int>& colors);
};

// distance-2 (column) colouring of `con`: greedy first-fit over the non-dominated rows.
// Returns number of colours.  Validity rule = reference DAColoring.C:931-1037.
int d2_coloring(const JacCon& con, std::vector<int>& colors);
bool validate_coloring(const JacCon& con, const std::vector<int>& colors);

}  // namespace das
