// app_common.hpp -- argv helpers shared by the three harness binaries. The binaries keep the
// reference's command lines and add --stencil=N: the n x n generator matrix
// built in memory instead of read from a .mtx file (a 20000^2 file is 48 GB of text).
#pragma once

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "spmv_amd.h"

namespace app {

inline const char* value_of(const char* arg, const char* key) {
    const size_t k = strlen(key);
    return strncmp(arg, key, k) == 0 ? arg + k : nullptr;
}

inline std::vector<std::string> split_modes(const char* text) {
    std::vector<std::string> out;
    std::string cur;
    for (const char* c = text; *c; ++c) {
        if (*c == ',') {
            if (!cur.empty()) out.push_back(cur);
            cur.clear();
        } else {
            cur.push_back(*c);
        }
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

// Generator matrix as COO in the writer's order (centre, left, right, top, bottom per grid point).
inline bool make_stencil(int n, MatrixData* mat) {
    const long long rows = (long long)n * n, nnz = 5LL * n * n - 4LL * n;
    if (n < 1 || rows > 0x7fffffffLL || nnz > 0x7fffffffLL) return false;
    Entry* e = (Entry*)malloc((size_t)nnz * sizeof(Entry));
    if (!e) return false;
    size_t k = 0;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            const int id = i * n + j;
            e[k++] = Entry{id, id, 5.0};
            if (j > 0) e[k++] = Entry{id, id - 1, -1.0};
            if (j < n - 1) e[k++] = Entry{id, id + 1, -1.0};
            if (i > 0) e[k++] = Entry{id, id - n, -1.0};
            if (i < n - 1) e[k++] = Entry{id, id + n, -1.0};
        }
    }
    mat->rows = mat->cols = (int)rows;
    mat->nnz = (int)nnz;
    mat->grid_size = n;
    mat->entries = e;
    return true;
}

// <base>_<opname>.<ext>, as main.cu:200-241 names the per-mode output files.
inline std::string per_mode_path(const char* path, const char* opname, const char* fallback_ext) {
    const char* dot = strrchr(path, '.');
    if (dot) return std::string(path, dot - path) + "_" + opname + dot;
    return std::string(path) + "_" + opname + fallback_ext;
}

}  // namespace app
