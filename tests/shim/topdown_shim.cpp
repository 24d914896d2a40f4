// topdown_shim.cpp — test infrastructure: the host half of the BVH builder (moonshine_amd/csrc/bvh_topdown.h) behind a C entry point, compiled with g++
// by tests/test_builder_host.py (no GPU): one top-down build over n boxes with primitive counts, leaves referenced as 0x80000000 | index.
#include "../../moonshine_amd/csrc/bvh_topdown.h"
#include <cstring>

extern "C" int topdown_build(const float* boxes /* 6 per element: lo xyz, hi xyz */, const uint32_t* counts, uint32_t n, uint32_t id_base,
                             uint32_t* left, uint32_t* right, float* node_boxes, float* cost7, uint8_t* split8 /* n - 1 entries each, by id - id_base */,
                             uint32_t* root_ref, uint32_t* deepest) {
    using namespace msne;
    if (n == 0) return -1;
    std::vector<TopCluster> el(n);
    for (uint32_t i = 0; i < n; i++) {
        el[i].ref = 0x80000000u | i; el[i].count = counts[i];
        for (int k = 0; k < 3; k++) { el[i].box.lo[k] = boxes[6 * i + k]; el[i].box.hi[k] = boxes[6 * i + 3 + k]; }
        for (int k = 0; k < 7; k++) el[i].cost[k] = 0.0f;
    }
    const uint32_t total = id_base + (n - 1);
    HostTree T;
    T.alloc(total); T.left.fill(0xFFFFFFFFu); T.right.fill(0xFFFFFFFFu); T.cost.fill(-1.0f); T.split.fill(0xFF);
    std::vector<uint32_t> ids(n - 1);
    for (uint32_t i = 0; i + 1 < n; i++) ids[i] = id_base + i;
    TopDown td(T, ids.data());
    const TopDown::Sub r = td.run(el.data(), n);
    if (td.used != n - 1) return -2;
    for (uint32_t i = 0; i + 1 < n; i++) {
        const uint32_t id = id_base + i;
        left[i] = T.left[id]; right[i] = T.right[id];
        for (int k = 0; k < 3; k++) { node_boxes[6 * i + k] = T.box[id].lo[k]; node_boxes[6 * i + 3 + k] = T.box[id].hi[k]; }
        std::memcpy(cost7 + 7 * (size_t)i, &T.cost[7 * (size_t)id], 28); std::memcpy(split8 + 8 * (size_t)i, &T.split[8 * (size_t)id], 8);
    }
    *root_ref = r.ref; *deepest = td.deepest;
    return 0;
}
