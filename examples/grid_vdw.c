/* Plain-C caller of libceg_hip.so: the VdW energy grid of a Lennard-Jones probe in a cubic cell with 32 atoms, through the same
 * one-shot entry point the Julia shim binds (ceg_grid_vdw, include/ceg_hip.h).  Shows that the boundary is a C ABI -- no C++, no
 * torch, no HIP types on the caller's side -- and is what tests/test_boundary_static.py compiles (gcc -std=c99) and, on a GPU box,
 * tests/test_gpu_parity.py runs and checks against the oracle.
 *
 *   gcc -std=c99 -Iinclude examples/grid_vdw.c -o /tmp/grid_vdw -Lcrystalenergygrids.jl_amd/csrc -lceg_hip \
 *       -Wl,-rpath,$PWD/crystalenergygrids.jl_amd/csrc && /tmp/grid_vdw out.f32
 *
 * Output: the raw float array [8][nx][ny][nz] (what create_grid_vdw keeps in memory, src/grids.jl:126-133) in `out.f32`,
 * and one line with a checksum. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "ceg_hip.h"

int main(int argc, char** argv)
{
    enum { NSIDE = 2, NATOMS = 4 * NSIDE * NSIDE * NSIDE, N = 23 };
    const double a = 26.0;                       /* cell edge, A: perpendicular widths >= 2 x the 12 A cutoff (src/probes.jl:24) */
    double pos[3 * NATOMS];
    int64_t kind[NATOMS];
    /* an fcc-like arrangement, two framework kinds */
    const double basis[4][3] = {{0.05, 0.05, 0.05}, {0.55, 0.55, 0.05}, {0.55, 0.05, 0.55}, {0.05, 0.55, 0.55}};
    int n = 0;
    for (int i = 0; i < NSIDE; ++i)
        for (int j = 0; j < NSIDE; ++j)
            for (int k = 0; k < NSIDE; ++k)
                for (int b = 0; b < 4; ++b, ++n) {
                    pos[3 * n] = (i + basis[b][0]) * a / NSIDE;
                    pos[3 * n + 1] = (j + basis[b][1]) * a / NSIDE;
                    pos[3 * n + 2] = (k + basis[b][2]) * a / NSIDE;
                    kind[n] = 1 + (b & 1);
                }
    const double mat[9] = {a, 0, 0, 0, a, 0, 0, 0, a};                         /* column-major */
    const double invmat[9] = {1 / a, 0, 0, 0, 1 / a, 0, 0, 0, 1 / a};
    /* ForceField column of the probe: kind 1 -> LJ(eps = 120 K, sigma = 3.4 A), kind 2 -> LJ(80 K, 3.0 A); params as the reference
     * stores them for FF.LennardJones (src/interactions.jl:367-372): p[0] = eps, p[1] = sigma; shift = the energy at the cutoff */
    ceg_rule_t rules[2] = {{CEG_LENNARDJONES, 0, {120.0, 3.4, 0.0}, 0.0}, {CEG_LENNARDJONES, 0, {80.0, 3.0, 0.0}, 0.0}};
    const int32_t rule_offset[3] = {0, 1, 2};
    const int32_t dims[3] = {N, N, N};
    const double size[3] = {a, a, a}, shift[3] = {0.0, 0.0, 0.0}, delta[3] = {a / N, a / N, a / N};
    const size_t count = (size_t)8 * (N + 1) * (N + 1) * (N + 1);
    float* grid = (float*)malloc(count * sizeof(float));
    if (!grid) return 2;
    if (ceg_device_count() <= 0) {
        fprintf(stderr, "no HIP device: %s\n", "this library has no CPU path");
        return 3;
    }
    const double GRID_TO_KELVIN = 0.01 * 120.27221933;        /* any positive scale does for the demonstration */
    const int rc = ceg_grid_vdw(pos, kind, NATOMS, mat, invmat, /*ortho*/ 1, /*safemin2*/ a * a / 4, /*cutoff2*/ 144.0,
                                rules, rule_offset, 2, dims, size, shift, delta, 1.0 / GRID_TO_KELVIN, GRID_TO_KELVIN * 1e7, grid, 1);
    if (rc != CEG_OK) {
        fprintf(stderr, "ceg_grid_vdw failed (%d): %s\n", rc, ceg_last_error());
        return 1;
    }
    double sum = 0.0;
    for (size_t t = 0; t < count; ++t) sum += grid[t] == grid[t] ? (double)grid[t] * (double)(1 + t % 7) : 0.0;
    printf("abi %d, %zu floats, weighted sum %.10e\n", ceg_abi_version(), count, sum);
    if (argc > 1) {
        FILE* f = fopen(argv[1], "wb");
        if (!f || fwrite(grid, sizeof(float), count, f) != count) return 4;
        fclose(f);
    }
    free(grid);
    return 0;
}
