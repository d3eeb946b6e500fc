// Host layer of the backend under AddressSanitizer + UBSan (the reference configures no sanitizer job; "new build:
// ASan-instrumented host build").  Exercises the two pure-host pieces of the C ABI — the ECS column store (world.cpp: spawn /
// insert / column views / rates / tick) and the commit hand-off (telemetry_sink.cpp: register / push / floor reads / series /
// commit_rows / copy_to_rows / truncate) — with growth, error paths and pointer invalidation the way a host uses them.
//   make -C elodin_amd/csrc asan      (g++ -fsanitize=address,undefined; no HIP, no GPU)      tests/test_asan_host.py runs it
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/sixdof_hip.h"

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "asan_host_test: CHECK failed at line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main() {
    // ---- world: 2,000 bodies spawned one by one (buffers reallocate many times), column views re-read after growth ----
    sixdof_world* w = sixdof_world_create();
    CHECK(w && sixdof_world_entity_len(w) == 1);                                   // entity 0 = Globals
    const uint64_t d7[1] = {7}, d6[1] = {6}, d3[1] = {3};
    for (int i = 0; i < 2000; i++) {
        const uint64_t e = sixdof_world_spawn(w);
        CHECK(e == static_cast<uint64_t>(i + 1));
        const double pos[7] = {0, 0, 0, 1, static_cast<double>(i), 0, 6}, vel[6] = {0, 0, 0, 1, 0, 0};
        CHECK(sixdof_world_insert(w, e, "world_pos", SIXDOF_PRIM_F64, d7, 1, pos, sizeof(pos)) == SIXDOF_OK);
        CHECK(sixdof_world_insert(w, e, "world_vel", SIXDOF_PRIM_F64, d6, 1, vel, sizeof(vel)) == SIXDOF_OK);
        if (i % 3 == 0) {                                                            // a component only some entities carry
            const float wind[3] = {0.5f, -1.0f, 0.0f};
            CHECK(sixdof_world_insert(w, e, "wind", SIXDOF_PRIM_F32, d3, 1, wind, sizeof(wind)) == SIXDOF_OK);
        }
    }
    const double bad[6] = {0};
    CHECK(sixdof_world_insert(w, 1, "world_pos", SIXDOF_PRIM_F64, d7, 1, bad, sizeof(bad)) != SIXDOF_OK);      // wrong byte count: refused
    CHECK(std::strlen(sixdof_world_last_error(w)) > 0);
    sixdof_column c;
    CHECK(sixdof_world_column(w, sixdof_component_id("world_pos"), &c) == SIXDOF_OK && c.n_rows == 2000 && c.dims[0] == 7);
    CHECK(static_cast<const double*>(c.host_ptr)[7 * 1999 + 4] == 1999.0 && c.entity_ids[1999] == 2000);
    CHECK(sixdof_world_column(w, sixdof_component_id("wind"), &c) == SIXDOF_OK && c.n_rows == 667 && c.prim_type == SIXDOF_PRIM_F32);
    CHECK(sixdof_world_column(w, sixdof_component_id("nobody"), &c) != SIXDOF_OK);
    uint64_t ids[16];
    const size_t n_comp = sixdof_world_components(w, ids, 16);
    CHECK(n_comp >= 5);
    for (size_t k = 1; k < n_comp && k < 16; k++) CHECK(ids[k - 1] < ids[k]);     // ascending ComponentId
    CHECK(sixdof_world_components(w, ids, 2) >= 5);                                  // a short buffer is not overrun
    CHECK(sixdof_world_set_rates(w, 120.0, 60.0) == SIXDOF_OK && sixdof_world_ticks_per_telemetry(w) == 2);
    CHECK(sixdof_world_time_step(w) == 0.008333333);
    CHECK(sixdof_world_set_rates(w, 120.0, 50.0) != SIXDOF_OK);                    // not an integer ratio
    CHECK(sixdof_world_set_rates(w, -1.0, 0.0) != SIXDOF_OK);
    sixdof_world_advance_tick(w, 5);
    CHECK(sixdof_world_tick(w) == 5);
    sixdof_world_destroy(w);

    // ---- sink: many pairs, many samples, floor reads, time travel, series views, commit / copy-back, truncate ----
    sixdof_sink* s = sixdof_sink_create();
    std::vector<uint64_t> pids;
    for (int e = 0; e < 64; e++) {
        char name[32];
        std::snprintf(name, sizeof(name), "body%d", e);
        const uint64_t pid = sixdof_pair_id(name, "world_pos");
        CHECK(sixdof_sink_register(s, pid, 56, name) == SIXDOF_OK);
        CHECK(sixdof_sink_register(s, pid, 56, name) == SIXDOF_OK);              // idempotent
        pids.push_back(pid);
    }
    CHECK(sixdof_sink_register(s, pids[0], 48, "body0") != SIXDOF_OK);             // same pair, another element size
    std::vector<double> rows(64 * 7);
    for (int t = 0; t < 500; t++) {
        for (size_t k = 0; k < rows.size(); k++) rows[k] = t + 0.001 * static_cast<double>(k);
        CHECK(sixdof_sink_commit_rows(s, pids.data(), rows.data(), 64, 56, 1000 * (t + 1)) == SIXDOF_OK);
    }
    CHECK(sixdof_sink_sample_count(s, pids[63]) == 500);
    double out[7];
    int64_t ts = 0;
    CHECK(sixdof_sink_latest(s, pids[5], &ts, out, 56) == SIXDOF_OK && ts == 500000 && out[0] == 499 + 0.001 * 35);
    CHECK(sixdof_sink_at(s, pids[5], 250500, &ts, out, 56) == SIXDOF_OK && ts == 250000);      // floor
    CHECK(sixdof_sink_at(s, pids[5], 10, &ts, out, 56) != SIXDOF_OK);                           // before the first sample
    CHECK(sixdof_sink_latest(s, pids[5], &ts, out, 48) != SIXDOF_OK);                           // wrong size
    CHECK(sixdof_sink_push(s, pids[5], 499999, out, 56) != SIXDOF_OK);                          // time travel
    CHECK(sixdof_sink_push(s, 12345, 1, out, 56) != SIXDOF_OK);                                 // unknown pair
    const int64_t* tsv = nullptr;
    const uint8_t* data = nullptr;
    uint64_t n = 0;
    uint32_t eb = 0;
    CHECK(sixdof_sink_series(s, pids[7], &tsv, &data, &n, &eb) == SIXDOF_OK && n == 500 && eb == 56 && tsv[499] == 500000);
    double last;
    std::memcpy(&last, data + 499 * 56, sizeof(double));
    CHECK(last == 499 + 0.001 * 49);
    std::vector<double> back(64 * 7, 0.0);
    int changed = 0;
    std::vector<uint64_t> some = pids;
    some[3] = 0;                                                                    // an entity without metadata: skipped
    CHECK(sixdof_sink_copy_to_rows(s, some.data(), back.data(), 64, 56, &changed) == SIXDOF_OK && changed == 1);
    CHECK(back[7 * 3] == 0.0 && back[7 * 4] == 499 + 0.001 * 28);
    CHECK(sixdof_sink_copy_to_rows(s, some.data(), back.data(), 64, 56, &changed) == SIXDOF_OK && changed == 0);
    uint64_t all[128];
    CHECK(sixdof_sink_pairs(s, all, 128) == 64 && sixdof_sink_pairs(s, all, 3) == 64);
    sixdof_sink_truncate(s);
    CHECK(sixdof_sink_sample_count(s, pids[0]) == 0 && sixdof_sink_push(s, pids[0], 1, out, 56) == SIXDOF_OK);
    sixdof_sink_destroy(s);
    std::printf("asan_host_test: ok\n");
    return 0;
}
