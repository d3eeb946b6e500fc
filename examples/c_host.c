/* c_host.c — a host in plain C over the C ABI (no Python): the shape of a `WorldExec::Hip` in another language.
 * Builds a three-body world with sixdof_world_*, binds it, runs RK4 ticks on the GPU, prints the poses.
 *   gcc -O2 -Iinclude examples/c_host.c -Lelodin_amd -lsixdof_hip -Wl,-rpath,$PWD/elodin_amd -o c_host && ./c_host 100
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sixdof_hip.h"

#define CHECK(call)                                                                     \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != SIXDOF_OK) {                                                         \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sixdof_last_error(h));        \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

static void insert(sixdof_world* w, uint64_t e, const char* name, const double* v, uint64_t n) {
    const uint64_t dims[2] = {n, 0};
    if (sixdof_world_insert(w, e, name, SIXDOF_PRIM_F64, dims, 1, v, n * sizeof(double)) != SIXDOF_OK) {
        fprintf(stderr, "insert %s: %s\n", name, sixdof_world_last_error(w));
        exit(1);
    }
}

int main(int argc, char** argv) {
    const uint64_t ticks = argc > 1 ? strtoull(argv[1], NULL, 10) : 100;
    const double G = 6.6743e-11, m = 1.0 / G;
    const double px[3] = {0.8920281421, -0.6628498947, -0.2291782474};
    const double vy[3] = {0.9957939373, -1.6191613336, 0.6233673964};
    sixdof_handle* h = NULL;
    sixdof_world* w = sixdof_world_create();
    uint64_t id[3];
    for (int k = 0; k < 3; k++) {
        const double pos[7] = {0, 0, 0, 1, px[k], 0, 0}, vel[6] = {0, 0, 0, 0, vy[k], 0}, zero[6] = {0};
        const double inertia[7] = {m, m, m, 0, 0, 0, m};
        id[k] = sixdof_world_spawn(w);
        insert(w, id[k], "world_pos", pos, 7);
        insert(w, id[k], "world_vel", vel, 6);
        insert(w, id[k], "world_accel", zero, 6);
        insert(w, id[k], "force", zero, 6);
        insert(w, id[k], "inertia", inertia, 7);
    }
    if (sixdof_world_set_rates(w, 120.0, 0.0) != SIXDOF_OK) return 1;

    sixdof_desc d;
    memset(&d, 0, sizeof d);
    d.struct_size = sizeof d;
    d.integrator = SIXDOF_INTEGRATOR_RK4;
    d.dtype = SIXDOF_F64;
    d.ticks_per_launch = 10;
    if (sixdof_create(&d, &h) != SIXDOF_OK) {
        fprintf(stderr, "sixdof_create: %s\n", sixdof_last_error(NULL));
        return 2;
    }
    CHECK(sixdof_bind_world(h, w));
    sixdof_effector_op op;
    memset(&op, 0, sizeof op);
    op.kind = SIXDOF_EFF_EDGE_GRAVITY_NEWTON;
    op.p[0] = G;
    CHECK(sixdof_set_effectors(h, &op, 1));
    /* examples/three-body/main.py:82-89 spawn order: a->b, b->a, a->c, b->c, c->a, c->b */
    const uint64_t from[6] = {id[0], id[1], id[0], id[1], id[2], id[2]};
    const uint64_t to[6] = {id[1], id[0], id[2], id[2], id[0], id[1]};
    CHECK(sixdof_set_edges(h, from, to, 6));
    CHECK(sixdof_upload(h));
    sixdof_timings t;
    CHECK(sixdof_step(h, ticks, &t));
    CHECK(sixdof_download(h, SIXDOF_COL_ALL));
    sixdof_world_advance_tick(w, ticks);

    sixdof_column c;
    sixdof_world_column(w, sixdof_component_id("world_pos"), &c);
    const double* pos = (const double*)c.host_ptr;
    for (uint64_t r = 0; r < c.n_rows; r++)
        printf("entity %llu  %.17g %.17g %.17g\n", (unsigned long long)c.entity_ids[r], pos[7 * r + 4], pos[7 * r + 5], pos[7 * r + 6]);
    printf("tick %llu, %llu launches, %.3f ms device\n", (unsigned long long)sixdof_world_tick(w),
           (unsigned long long)t.launches, t.kernel_device_ms);
    sixdof_destroy(h);
    sixdof_world_destroy(w);
    return 0;
}
