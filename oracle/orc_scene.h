/*
 * ORACLE — test infrastructure only (see orc_math.h header).
 * orc_scene.h: host-side scene tables of the reference, restated as plain C structs.
 *   shaders/hrtsystem/world.hlsl:6-72, engine/hrtsystem/{Accel,MeshManager,MaterialManager,BackgroundManager}.zig
 */
#ifndef ORC_SCENE_H
#define ORC_SCENE_H

#include "orc_math.h"
#include "../include/moonshine_amd.h"

typedef struct { float *rgba; uint32_t w, h; } orc_texture;             /* decoded to float RGBA */

typedef struct {                                                        /* MeshManager.zig:17-32 */
    v3 *positions; v3 *normals; v2 *texcoords; uint32_t *indices;       /* indices: 3 per triangle */
    uint32_t position_count, attribute_count, index_count;
} orc_mesh;

typedef struct { uint32_t normal, emissive, type, color, metalness, roughness; float ior; } orc_material;

typedef struct { uint32_t mesh, material, sampled; } orc_geometry;      /* world.hlsl:19-23 */

typedef struct {
    m34 transform, world_to_instance;
    int visible;
    uint32_t geo_offset, geo_count;     /* instanceID() = geo_offset (Accel.zig:394-412) */
    uint32_t blas;
} orc_instance;

/* canonical BVH: LBVH, 30-bit Morton, leaves <=4, collapsed to <=8-wide */
typedef struct { v3 v0, v1, v2; uint32_t geo, prim; } orc_tri;          /* object space */
typedef struct {
    float lo[3][8], hi[3][8];   /* [axis][child]: the eight box tests of a visit are one loop the compiler vectorises (slots >= nchild hold zeros and are ignored) */
    float grow[8];              /* 1e-4 of the child box's largest extent (0 for an empty or unbounded box): orc_bvh.c box_hit8 */
    uint32_t child[8];      /* internal: node index; leaf: first item */
    uint8_t  count[8];      /* 0 = internal, else number of items in the leaf */
    uint8_t  nchild;
} orc_wnode;
typedef struct {
    orc_wnode *nodes; uint32_t node_count;
    orc_tri *tris; uint32_t tri_count;          /* BLAS items */
    uint32_t *inst; uint32_t inst_count;        /* TLAS items (instance indices) */
    v3 lo, hi;
    float cull_far;                             /* TLAS: orc_bvh.c instance_cull_slack, the largest per-unit-of-origin slack of any instance */
} orc_bvh;

typedef struct { uint32_t inst, geo, prim; float t, u, v; } orc_hit;                 /* intersection.hlsl:5-9; inst==MAX_UINT: miss */

typedef struct { uint32_t alias; float select; uint32_t instance, geometry, primitive; } orc_alias_entry; /* light.hlsl:17-22,112-116 */

typedef struct {
    uint32_t size;              /* S */
    float *rgb;                 /* S*S*4 (equal-area map, RGBA32F) */
    float **lum; uint32_t mip_count; /* lum[l] has (S>>l)^2 texels */
} orc_envmap;

typedef struct { float *film; uint32_t w, h, sample_count; } orc_sensor;

typedef struct {
    uint64_t closest_rays, shadow_rays, samples, surface_hits;
    uint64_t node_visits, tri_tests;                 /* all rays */
    uint64_t shadow_node_visits, shadow_tri_tests;   /* any-hit rays only */
} orc_counters;

typedef struct OrcContext {
    orc_texture *textures; uint32_t texture_count;
    orc_mesh *meshes; uint32_t mesh_count;
    orc_material *materials; uint32_t material_count;
    orc_geometry *geometries; uint32_t geometry_count;
    orc_instance *instances; uint32_t instance_count;
    orc_bvh *blases; uint32_t blas_count; uint32_t *blas_key_off, *blas_key_len; uint32_t *blas_keys; uint32_t blas_keys_len;
    orc_bvh tlas; int accel_dirty;
    int exhaustive;         /* OrcSetExhaustiveSearch: 0 = both levels culled by their boxes; 1 = every visible instance is entered; 2 = and every triangle of it is tested */
    orc_alias_entry *alias; /* entry 0 = header */
    orc_envmap env;
    orc_sensor *sensors; uint32_t sensor_count;
    Lens *lenses; uint32_t lens_count;
    MsnePipelineOpts opts;
    uint32_t tile_size, shard_index, shard_count;
    int threads;
    orc_counters counters;
    float srgb_lut[256];
} OrcContext;

#endif
