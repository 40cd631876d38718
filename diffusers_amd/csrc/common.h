// common.h — shared declarations of the engine (host side).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sdengine.h"

typedef _Float16 half_t;

struct SdeError : std::runtime_error {
    sde_status code;
    SdeError(sde_status c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void sde_fail(sde_status c, const std::string& m) { throw SdeError(c, m); }

#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            sde_fail(SDE_ERR_DEVICE, std::string(#expr) + " failed: " + hipGetErrorString(_e) +      \
                                         " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")");    \
    } while (0)

#define SDE_REQUIRE(cond, code, msg)                \
    do {                                            \
        if (!(cond)) sde_fail((code), (msg));       \
    } while (0)

void sde_set_last_error(const std::string& m);

// every C ABI entry point runs inside this: nothing is thrown across the boundary
template <typename F>
static inline sde_status guarded(F&& f)
{
    try {
        f();
        return SDE_OK;
    } catch (const SdeError& e) {
        sde_set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        sde_set_last_error("out of host memory");
        return SDE_ERR_DEVICE;
    } catch (const std::exception& e) {
        sde_set_last_error(e.what());
        return SDE_ERR_INVALID_ARGUMENT;
    } catch (...) {
        sde_set_last_error("unknown error");
        return SDE_ERR_INVALID_ARGUMENT;
    }
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }
