// world.cpp — host-side ECS column store, the C++ counterpart of nox-py's `World`
// (libs/nox-py/src/world.rs:23-45,174-229,238-276): per component one growing row-major byte buffer plus the
// entity id of every row, rows in spawn order, ids handed out sequentially, entity 0 = "Globals" carrying
// `tick` (u64) and `simulation_time_step` (f64).  Pure host code (no HIP): usable and tested without a GPU.
// A world is bound to a backend handle with sixdof_bind_world (sixdof_capi.cpp).
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/sixdof_hip.h"
#include "abi_guard.hpp"

struct WorldColumn {
    std::string name;
    int prim = SIXDOF_PRIM_F64;
    uint32_t ndim = 0;
    uint64_t dims[2] = {0, 0};
    size_t row_bytes = 0;
    std::vector<uint8_t> buffer;       // Column.buffer
    std::vector<uint64_t> entity_ids;  // Column.entity_ids
};

struct sixdof_world {
    std::map<uint64_t, WorldColumn> host;   // BTreeMap<ComponentId, Column>: ascending id
    uint64_t entity_len = 0;                // metadata.entity_len
    uint64_t tick = 0;                      // metadata.tick
    double sim_time_step = 0.0;
    uint64_t ticks_per_telemetry = 1;
    std::string err;
};

static size_t prim_size(int prim) { return prim == SIXDOF_PRIM_F32 ? 4 : 8; }
static std::string* err_of(const sixdof_world* w) { return w ? &const_cast<sixdof_world*>(w)->err : nullptr; }

extern "C" {

// ComponentId::new: FNV-1a-64 & !(1 << 63) (impeller2/src/types.rs:39-44, const-fnv1a-hash)
uint64_t sixdof_component_id(const char* name) {
    uint64_t h = 0xcbf29ce484222325ull;
    if (name)
        for (const unsigned char* p = reinterpret_cast<const unsigned char*>(name); *p; ++p) {
            h ^= static_cast<uint64_t>(*p);
            h *= 0x100000001b3ull;
        }
    return h & ~(1ull << 63);
}

// Duration::from_secs_f64(1/rate).as_secs_f64(): ns-quantised (world_builder.rs:221, world.rs:185-191)
double sixdof_quantize_time_step(double rate_hz) {
    if (!(rate_hz > 0.0)) return std::nan("");
    const long double ns = nearbyintl(static_cast<long double>(1.0 / rate_hz) * 1.0e9L);
    const uint64_t total = static_cast<uint64_t>(ns);
    return static_cast<double>(total / 1000000000ull) + static_cast<double>(total % 1000000000ull) / 1.0e9;
}

sixdof_world* sixdof_world_create(void) try {
    std::unique_ptr<sixdof_world> owner(new sixdof_world());      // released only once the globals are in
    sixdof_world* w = owner.get();
    // add_globals (world.rs:174-183): SystemGlobals::new(sim_time_step) on entity 0; DEFAULT_TIME_STEP = 1/120 s in ns
    const uint64_t globals = w->entity_len++;
    const uint64_t tick0 = 0;
    w->sim_time_step = static_cast<double>(1000000000ull / 120) / 1.0e9;
    if (sixdof_world_insert(w, globals, "tick", SIXDOF_PRIM_U64, nullptr, 0, &tick0, 8) != SIXDOF_OK) return nullptr;
    if (sixdof_world_insert(w, globals, "simulation_time_step", SIXDOF_PRIM_F64, nullptr, 0, &w->sim_time_step, 8) != SIXDOF_OK) return nullptr;
    return owner.release();
} SIXDOF_ABI_CATCH_VALUE(nullptr, nullptr)

void sixdof_world_destroy(sixdof_world* w) { delete w; }

const char* sixdof_world_last_error(const sixdof_world* w) { return w ? w->err.c_str() : ""; }

uint64_t sixdof_world_spawn(sixdof_world* w) { return w ? w->entity_len++ : 0; }

uint64_t sixdof_world_entity_len(const sixdof_world* w) { return w ? w->entity_len : 0; }

int sixdof_world_insert(sixdof_world* w, uint64_t entity, const char* component, int prim, const uint64_t* dims,
                        uint32_t ndim, const void* row, size_t n_bytes) try {
    if (!w || !component || (!row && n_bytes) || ndim > 2 || (ndim && !dims)) return SIXDOF_ERR_INVALID_ARGUMENT;
    if (entity >= w->entity_len) {
        w->err = "insert: unknown entity";
        return SIXDOF_ERR_INVALID_ARGUMENT;
    }
    size_t elems = 1;
    for (uint32_t k = 0; k < ndim; k++) elems *= dims[k];
    if (elems * prim_size(prim) != n_bytes) {
        w->err = std::string("insert: row of component ") + component + " has the wrong byte size";
        return SIXDOF_ERR_VALUE_SIZE_MISMATCH;
    }
    const uint64_t id = sixdof_component_id(component);
    auto it = w->host.find(id);
    const bool fresh = it == w->host.end();
    if (fresh) {
        WorldColumn c;
        c.name = component;
        c.prim = prim;
        c.ndim = ndim;
        for (uint32_t k = 0; k < ndim; k++) c.dims[k] = dims[k];
        c.row_bytes = n_bytes;
        it = w->host.emplace(id, std::move(c)).first;
    } else if (it->second.row_bytes != n_bytes || it->second.prim != prim) {
        w->err = std::string("insert: value size mismatch on component ") + component;   // Error::ValueSizeMismatch
        return SIXDOF_ERR_VALUE_SIZE_MISMATCH;
    }
    WorldColumn& c = it->second;
    const uint8_t* p = static_cast<const uint8_t*>(row);
    try {       // whatever can throw throws before a byte changes: a failed insert leaves the world as it was
        c.entity_ids.reserve(c.entity_ids.size() + 1);
        c.buffer.insert(c.buffer.end(), p, p + n_bytes);
    } catch (...) {
        if (fresh) w->host.erase(it);
        throw;      // -> SIXDOF_ERR_OUT_OF_MEMORY at the barrier below
    }
    c.entity_ids.push_back(entity);
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(w))

int sixdof_world_column(sixdof_world* w, uint64_t component_id, sixdof_column* out) try {
    if (!w || !out) return SIXDOF_ERR_INVALID_ARGUMENT;
    auto it = w->host.find(component_id);
    if (it == w->host.end()) {
        w->err = "column: component not found";
        return SIXDOF_ERR_COMPONENT_NOT_FOUND;   // Error::ComponentNotFound
    }
    WorldColumn& c = it->second;
    out->component_id = component_id;
    out->prim_type = c.prim;
    out->ndim = c.ndim;
    out->dims[0] = c.dims[0];
    out->dims[1] = c.dims[1];
    out->n_rows = c.entity_ids.size();            // len = bytes / size (world.rs:335-337)
    out->entity_ids = c.entity_ids.data();
    out->host_ptr = c.buffer.data();
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(w))

size_t sixdof_world_components(const sixdof_world* w, uint64_t* ids, size_t cap) try {
    if (!w) return 0;
    size_t k = 0;
    for (auto& kv : w->host) {     // ascending ComponentId, like the reference's BTreeMap
        if (ids && k < cap) ids[k] = kv.first;
        k++;
    }
    return k;
} SIXDOF_ABI_CATCH_VALUE(err_of(w), 0)

// validate_rates + set_globals (world_builder.rs:211-243, world.rs:185-191)
int sixdof_world_set_rates(sixdof_world* w, double simulation_rate_hz, double telemetry_rate_hz) try {
    if (!w) return SIXDOF_ERR_INVALID_ARGUMENT;
    if (!(simulation_rate_hz > 0.0)) {
        w->err = "simulation_rate must be > 0 Hz, got " + std::to_string(simulation_rate_hz);
        return SIXDOF_ERR_INVALID_ARGUMENT;
    }
    uint64_t tpt = 1;
    if (telemetry_rate_hz != 0.0) {
        const double ratio = simulation_rate_hz / telemetry_rate_hz;
        const double rounded = std::round(ratio);
        if (!(telemetry_rate_hz > 0.0) || std::fabs(ratio - rounded) > 1e-9 || rounded < 1.0) {
            w->err = "telemetry_rate (" + std::to_string(telemetry_rate_hz) + " Hz) must evenly divide simulation_rate (" +
                     std::to_string(simulation_rate_hz) + " Hz); got ratio " + std::to_string(ratio);
            return SIXDOF_ERR_INVALID_ARGUMENT;
        }
        tpt = static_cast<uint64_t>(rounded);
    }
    w->sim_time_step = sixdof_quantize_time_step(simulation_rate_hz);
    w->ticks_per_telemetry = tpt ? tpt : 1;
    auto it = w->host.find(sixdof_component_id("simulation_time_step"));
    if (it != w->host.end() && it->second.buffer.size() >= 8) std::memcpy(it->second.buffer.data(), &w->sim_time_step, 8);
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(w))

double sixdof_world_time_step(const sixdof_world* w) { return w ? w->sim_time_step : 0.0; }
uint64_t sixdof_world_ticks_per_telemetry(const sixdof_world* w) { return w ? w->ticks_per_telemetry : 1; }
uint64_t sixdof_world_tick(const sixdof_world* w) { return w ? w->tick : 0; }

// advance_tick (world.rs:276-278) + the globals `tick` column the compiled tick increments (globals.rs:42-44)
void sixdof_world_advance_tick(sixdof_world* w, uint64_t n) try {
    if (!w) return;
    w->tick += n;
    auto it = w->host.find(sixdof_component_id("tick"));
    if (it != w->host.end() && it->second.buffer.size() >= 8) std::memcpy(it->second.buffer.data(), &w->tick, 8);
} SIXDOF_ABI_CATCH_VALUE(err_of(w), )

}  // extern "C"
