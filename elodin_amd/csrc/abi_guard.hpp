// abi_guard.hpp — the exception barrier of the C ABI.
//
// The callers of include/sixdof_hip.h are `extern "C"` bindings from Rust (the reference side is
// libs/nox-py/src/cranelift_exec.rs:11,163-165) and ctypes: a C++ exception that unwinds through such a frame is undefined
// behaviour.  The reference reports failures as an enum (libs/nox-py/src/error.rs:12-51), so every entry point that can
// allocate (std::vector / std::string / std::map / new) is a function-try-block closed by one of the macros below: the
// exception is classified once, a message is left where `sixdof_*_last_error` finds it, and a status (or the type's neutral
// value for the few entry points that return no status) goes back to the caller.
//
//   int sixdof_foo(sixdof_handle* h, ...) try {
//       ...
//   } SIXDOF_ABI_CATCH(err_of(h))
//
// SIXDOF_ERR_OUT_OF_MEMORY: std::bad_alloc, and std::length_error (a container asked for more than max_size(): the same
// caller mistake — an absurd row count — one allocation earlier).  Anything else: SIXDOF_ERR_INTERNAL.
#pragma once

#include <cstdio>
#include <exception>
#include <new>
#include <stdexcept>
#include <string>

#include "../../include/sixdof_hip.h"

namespace sixdof_abi {

// Lippincott function: call only from inside a catch block.  Never throws (the message itself may fail to allocate: the
// fixed text of the out-of-memory case fits the small-string buffer, anything longer is dropped).
inline int caught(std::string* err, const char* fn) noexcept {
    int rc = SIXDOF_ERR_INTERNAL;
    char msg[192];
    try {
        throw;
    } catch (const std::bad_alloc&) {
        rc = SIXDOF_ERR_OUT_OF_MEMORY;
        std::snprintf(msg, sizeof msg, "%s: out of memory (std::bad_alloc)", fn);
    } catch (const std::length_error& e) {
        rc = SIXDOF_ERR_OUT_OF_MEMORY;
        std::snprintf(msg, sizeof msg, "%s: out of memory (std::length_error: %s)", fn, e.what());
    } catch (const std::exception& e) {
        std::snprintf(msg, sizeof msg, "%s: internal error (%s)", fn, e.what());
    } catch (...) {
        std::snprintf(msg, sizeof msg, "%s: internal error (unknown exception)", fn);
    }
    if (err) {
        try {
            err->assign(msg);
        } catch (...) {
            try {
                err->assign("out of memory");      // 13 characters: no allocation
            } catch (...) {
            }
        }
    }
    return rc;
}

}  // namespace sixdof_abi

// closes a function-try-block of an entry point that returns a sixdof_status
#define SIXDOF_ABI_CATCH(err_ptr) \
    catch (...) { return ::sixdof_abi::caught((err_ptr), __func__); }
// ... of an entry point that returns no status: the message is recorded, `neutral` (nullptr / 0 / nothing) is returned
#define SIXDOF_ABI_CATCH_VALUE(err_ptr, neutral) \
    catch (...) { ::sixdof_abi::caught((err_ptr), __func__); return neutral; }
