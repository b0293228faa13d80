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
