// Tiny assertion helpers for the C++ API tests (no test framework in the image).
#pragma once
#include <cstdio>
#include <cstdlib>

static int g_checks = 0, g_failures = 0;

#define REQUIRE(cond)                                                                              \
    do {                                                                                           \
        g_checks++;                                                                                \
        if (!(cond)) {                                                                             \
            g_failures++;                                                                          \
            if (g_failures <= 20)                                                                  \
                std::fprintf(stderr, "%s:%d: REQUIRE(%s) failed\n", __FILE__, __LINE__, #cond);    \
        }                                                                                          \
    } while (0)

#define TEST_CASE(name) static void name()

inline int finish(const char *suite) {
    std::printf("%s: %d checks, %d failures\n", suite, g_checks, g_failures);
    return g_failures == 0 ? 0 : 1;
}
