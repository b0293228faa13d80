// Launch geometry shared by the HBM-bound NHWC kernels: a tensor is a [rows, C] matrix with row
// stride ld; a thread owns one float4 channel group (c4) and walks rows.  blockDim = (cx, ry) with
// cx = power of two >= min(C/4, 64) so a wave reads whole 256 B..1 KiB runs of a pixel row.
#pragma once
#include "segmi_common.h"

struct RowGeom {
    int cx, ry;   // threads along channels / rows (cx*ry == 256)
    int c4;       // float4 groups per row (ceil(C/4))
    dim3 grid, block;
};

static inline RowGeom row_geom(long rows, int C, int rows_per_thread, int max_grid_y) {
    RowGeom g;
    g.c4 = (C + 3) / 4;
    int cx = 1;
    while (cx < g.c4 && cx < 64) cx <<= 1;
    g.cx = cx;
    g.ry = 256 / cx;
    long gy = (rows + (long)g.ry * rows_per_thread - 1) / ((long)g.ry * rows_per_thread);
    if (gy < 1) gy = 1;
    if (gy > max_grid_y) gy = max_grid_y;
    g.grid = dim3((unsigned)((g.c4 + cx - 1) / cx), (unsigned)gy);
    g.block = dim3((unsigned)cx, (unsigned)g.ry);
    return g;
}

// The same for ELEMENTWISE kernels (no block reduction over the thread rows, so ry need not be a power of two): a channel count whose
// float4 groups are not a power of two — 150 classes = 38 groups, 19 / 21 classes = 5 / 6 — gets cx = c4 exactly instead of the next
// power of two (38 of 64 lanes were live in the bilinear kernels at cfg5).  With ld == round_up(C, 4) consecutive rows are contiguous
// in memory, so a wave still touches one contiguous run.
static inline RowGeom row_geom_dense(long rows, int C, int rows_per_thread, int max_grid_y) {
    RowGeom g = row_geom(rows, C, rows_per_thread, max_grid_y);
    if (g.c4 >= 64 || (g.c4 & (g.c4 - 1)) == 0) return g;
    g.cx = g.c4;
    g.ry = 256 / g.cx;
    long gy = (rows + (long)g.ry * rows_per_thread - 1) / ((long)g.ry * rows_per_thread);
    if (gy < 1) gy = 1;
    if (gy > max_grid_y) gy = max_grid_y;
    g.grid = dim3(1u, (unsigned)gy);
    g.block = dim3((unsigned)g.cx, (unsigned)g.ry);
    return g;
}
