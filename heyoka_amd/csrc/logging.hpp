// Minimal stage logger: decomposition sizes, the planner's verdict (and WHY a decomposition fell back to another
// generator), code-generation and hiprtc times.
//
// Reference: include/heyoka/logging.hpp:19-24 (set_logger_level_trace / debug / info / warn / err / critical),
// src/logging.cpp:20-48, and the stopwatches of the construction path - src/taylor_01.cpp:321 / :439-440 (CSE),
// :642 (topological sort), :968 (decomposition), src/taylor_adaptive_batch.cpp:322, :348 (dense output, compilation).
// The reference logs through spdlog (default level: warn); here a few lines to stderr - or to a callback installed
// through the C ABI (hy_set_log_callback) - at the same levels and with the same "runtime" wording.
#pragma once

#include <chrono>
#include <string>

namespace heyoka_amd
{

enum class log_level : int { trace = 0, debug = 1, info = 2, warn = 3, err = 4, critical = 5, off = 6 };

void set_logger_level(log_level);
log_level get_logger_level();
void set_logger_level_trace();
void set_logger_level_debug();
void set_logger_level_info();
void set_logger_level_warn();
void set_logger_level_err();
void set_logger_level_critical();

// Sink: nullptr = stderr ("[heyoka_amd] [level] message").
using log_sink_t = void (*)(int level, const char *msg, void *user);
void set_log_sink(log_sink_t, void *user);

namespace detail
{

bool log_enabled(log_level);
void log_message(log_level, const std::string &);

// Stopwatch in the spirit of spdlog::stopwatch (seconds, printed with 6 digits).
struct stopwatch {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double elapsed() const
    {
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    std::string str() const;
};

} // namespace detail

} // namespace heyoka_amd
