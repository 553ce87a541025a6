/* A C11 (not C++) caller of the reference's device API as a program built against kernel.h
 * would use it (driver.c:110,135): includes the library's header, builds a Triangle[] and the
 * 1-indexed vertex arrays, transforms them with mat_mult_device and intersects two rays with
 * intersect_all_triangles_device.  Prints the results in hex floats; tests/test_c_boundary.py
 * compares them with the CPU oracle's. */
#include <stdio.h>
#include <string.h>
#include "rt_amd.h"

_Static_assert(sizeof(Triangle) == 488, "Triangle layout as the reference's (LP64, _MAX_FNAME 256)");

int main(void)
{
    /* two triangles facing the origin, the second one behind the first and larger */
    double x[7] = {0, -1, 1, 0, -3, 3, 0};
    double y[7] = {0, -1, -1, 1, -3, -3, 3};
    double z[7] = {0, 5, 5, 5, 9, 9, 9};
    Triangle tris[2];
    memset(tris, 0, sizeof tris);
    tris[0].A = 1; tris[0].B = 2; tris[0].C = 3; tris[0].mtl.index = -1;
    tris[1].A = 4; tris[1].B = 5; tris[1].C = 6; tris[1].mtl.index = -1;
    /* a shear-free scale + translation, rows 0..2 used (mat_mult.cu:8-28) */
    double m[4][4] = {{1.5, 0, 0, 0.25}, {0, 1.5, 0, -0.125}, {0, 0, 2, 1}, {0, 0, 0, 1}};
    mat_mult_device(x, y, z, m, x, y, z, 7);
    for (int i = 0; i < 7; ++i) printf("v %d %a %a %a\n", i, x[i], y[i], z[i]);

    double S[3] = {0, 0, 0};
    double rays[3][3] = {{0.02, 0.01, 1}, {0.15, -0.1, 1}, {5, 5, 1}};   /* first, second, miss */
    for (int r = 0; r < 3; ++r) {
        double uv[2] = {-7, -7}, point[3] = {-7, -7, -7};
        int hit = intersect_all_triangles_device(S, rays[r], uv, point, tris, x, y, z, 2);
        printf("r %d %d %a %a %a %a %a\n", r, hit, uv[0], uv[1], point[0], point[1], point[2]);
    }
    return 0;
}
