// rg_abi_guard.h -- nothing C++ may leave the C ABI (a Rust / C / ctypes caller cannot unwind through it).
// Every `extern "C" int` entry point of csrc/abi_*.hip is a function-try-block that ends in RG_ABI_GUARD: a host allocation that
// fails (the mirror's tables, a result vector) comes back as RG_ERR_OUT_OF_MEMORY like a failed hipMalloc does, anything else as
// RG_ERR_STATE, the text through rg_last_error() as always. rg_fail itself allocates nothing (a fixed thread-local buffer).
// No HIP in this header: tests/test_abi.py compiles it with g++ and throws through it.
#pragma once
#include <exception>
#include <new>

#include "../../include/raftgroups.h"

int rg_fail(int code, const char *fmt, ...); // abi_state.hip: records the text, returns `code`

#define RG_ABI_GUARD                                                                                                     \
    catch (const std::bad_alloc &) { return rg_fail(RG_ERR_OUT_OF_MEMORY, "host memory allocation failed"); }           \
    catch (const std::exception &e__) { return rg_fail(RG_ERR_STATE, "unexpected C++ exception: %s", e__.what()); }      \
    catch (...) { return rg_fail(RG_ERR_STATE, "unexpected C++ exception"); }
