// See logging.hpp.
#include "logging.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>

namespace heyoka_amd
{

namespace
{

std::atomic<int> &level_store()
{
    // Default: warn, like the reference's logger; HEYOKA_AMD_LOG_LEVEL=trace|debug|info|warn|err|critical|off presets it.
    static std::atomic<int> lvl = [] {
        int l = static_cast<int>(log_level::warn);
        if (const char *e = std::getenv("HEYOKA_AMD_LOG_LEVEL")) {
            const std::string s(e);
            const char *names[] = {"trace", "debug", "info", "warn", "err", "critical", "off"};
            for (int i = 0; i < 7; ++i) {
                if (s == names[i]) {
                    l = i;
                }
            }
        }
        return l;
    }();
    return lvl;
}

std::mutex sink_mutex;
log_sink_t sink_fn = nullptr;
void *sink_user = nullptr;

} // namespace

void set_logger_level(log_level l)
{
    level_store().store(static_cast<int>(l));
}
log_level get_logger_level()
{
    return static_cast<log_level>(level_store().load());
}
void set_logger_level_trace()
{
    set_logger_level(log_level::trace);
}
void set_logger_level_debug()
{
    set_logger_level(log_level::debug);
}
void set_logger_level_info()
{
    set_logger_level(log_level::info);
}
void set_logger_level_warn()
{
    set_logger_level(log_level::warn);
}
void set_logger_level_err()
{
    set_logger_level(log_level::err);
}
void set_logger_level_critical()
{
    set_logger_level(log_level::critical);
}

void set_log_sink(log_sink_t fn, void *user)
{
    const std::lock_guard<std::mutex> lock(sink_mutex);
    sink_fn = fn;
    sink_user = user;
}

namespace detail
{

bool log_enabled(log_level l)
{
    return static_cast<int>(l) >= level_store().load();
}

void log_message(log_level l, const std::string &msg)
{
    if (!log_enabled(l)) {
        return;
    }
    static const char *names[] = {"trace", "debug", "info", "warning", "error", "critical", "off"};
    const std::lock_guard<std::mutex> lock(sink_mutex);
    if (sink_fn != nullptr) {
        sink_fn(static_cast<int>(l), msg.c_str(), sink_user);
    } else {
        std::fprintf(stderr, "[heyoka_amd] [%s] %s\n", names[static_cast<int>(l)], msg.c_str());
    }
}

std::string stopwatch::str() const
{
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%.6f", elapsed());
    return buf;
}

} // namespace detail

} // namespace heyoka_amd
