// rtiow/ProfileLogScope.hpp — RAII wall-clock scope that reports to stderr on exit
// (reference: ProfileLogScope.hpp:12-36, ProfileLogScope.cpp:5-15).
#pragma once

#include <chrono>
#include <iostream>

class ProfileLogScope {
public:
    explicit ProfileLogScope(const char* scopeName) noexcept : _scope_name(scopeName), _time_at_creation(std::chrono::steady_clock::now()) {}
    ~ProfileLogScope() noexcept {
        const std::chrono::duration<float> elapsed = std::chrono::steady_clock::now() - _time_at_creation;
        std::cerr << "ProfileLogScope " << _scope_name << " took " << elapsed.count() << " seconds.\n";
    }
    ProfileLogScope() = delete;
    ProfileLogScope(const ProfileLogScope&) = delete;
    ProfileLogScope& operator=(const ProfileLogScope&) = delete;

private:
    const char* _scope_name = nullptr;
    std::chrono::time_point<std::chrono::steady_clock> _time_at_creation{};
};

#define RTIOW_PASTE2(x, y) x##y
#define RTIOW_PASTE(x, y) RTIOW_PASTE2(x, y)
#undef PROFILE_LOG_SCOPE
#undef PROFILE_LOG_SCOPE_FUNCTION
#define PROFILE_LOG_SCOPE(tag_str) ProfileLogScope RTIOW_PASTE(plscope_, __LINE__)(tag_str)
// The reference expands this to MSVC's __FUNCSIG__; __PRETTY_FUNCTION__ is the GCC/Clang spelling.
#define PROFILE_LOG_SCOPE_FUNCTION() PROFILE_LOG_SCOPE(__PRETTY_FUNCTION__)
