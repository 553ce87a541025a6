/*
 * rt_reference_abi.h -- binary layout of the two aggregates that cross the reference's
 * device API (kernel.h:16-17 takes a `Triangle*`).
 *
 * The reference declares them in obj_reader.h:7-39 in terms of SDL_Surface* and the MSVC
 * constant _MAX_FNAME (256 on that platform; 260 is its _MAX_PATH), which gives
 * sizeof(Triangle) = 488 with 8-byte pointers.  A drop-in library cannot include SDL, so the same layout is
 * spelled out here with an opaque surface pointer and a compile-time name length.  Code
 * built against the reference's own obj_reader.h can pass its arrays straight through as
 * long as RT_REF_MAX_FNAME equals the _MAX_FNAME it was built with; the library reads only
 * the nine indices and mtl.index/Ka/Kd/Ks, at offsets computed from these definitions, and
 * takes the array stride from rt_set_triangle_stride() when a caller's build differs.
 */
#ifndef RT_REFERENCE_ABI_H
#define RT_REFERENCE_ABI_H

#include <stddef.h>

#ifndef RT_REF_MAX_FNAME
#define RT_REF_MAX_FNAME 256
#endif

#ifdef __cplusplus
extern "C" {
#endif

struct SDL_Surface;                      /* opaque; never dereferenced by this library */

typedef struct Material {                /* obj_reader.h:7-23 */
    int index;                           /* position in mats[], -1 = "no usemtl seen" */
    char name[RT_REF_MAX_FNAME];
    double Ka[3];
    double Kd[3];
    double Ks[3];
    double Ke[3];
    int illum;
    double Ns;
    double Tr;
    double Tf[3];
    double Ni;
    double d;
    struct SDL_Surface* map_Ka;
    struct SDL_Surface* map_Kd;
    struct SDL_Surface* map_Ks;
} Material;

typedef struct Triangle {                /* obj_reader.h:25-39 */
    int A, B, C;                         /* vertex indices (1-based into x,y,z)        */
    int At, Bt, Ct;                      /* texture-vertex indices (1-based into u,v)  */
    int An, Bn, Cn;                      /* normal indices (1-based into x/y/znormal)  */
    Material mtl;                        /* per-triangle COPY of the material          */
} Triangle;

#ifdef __cplusplus
}
#define RT_ABI_STATIC_ASSERT(cond, msg) static_assert(cond, msg)
#else
#define RT_ABI_STATIC_ASSERT(cond, msg) _Static_assert(cond, msg)
#endif
/* checked in C and C++ translation units alike */
RT_ABI_STATIC_ASSERT(RT_REF_MAX_FNAME != 256 || sizeof(Triangle) == 488,
                     "layout drifted from the reference's 488-byte Triangle");
RT_ABI_STATIC_ASSERT(offsetof(Triangle, mtl) == 40, "mtl must start at byte 40");
RT_ABI_STATIC_ASSERT(RT_REF_MAX_FNAME != 256 || offsetof(Triangle, mtl.Ka) == 304,
                     "Ka must start at byte 304");

#endif /* RT_REFERENCE_ABI_H */
