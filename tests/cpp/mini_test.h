// mini_test.h -- a tiny self-registering test harness (the reference uses gtest 1.6, which is not
// part of this repository).  TEST / TEST_F / EXPECT_* / ASSERT_* cover what the reference's
// test_pdfz*.cpp use, with the same meaning (ASSERT_FLOAT_EQ = within 4 float ulps).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

namespace mini {
struct Failure {
  std::string what;
};
struct Case {
  const char* suite;
  const char* name;
  void (*fn)();
};
inline std::vector<Case>& registry() {
  static std::vector<Case> r;
  return r;
}
struct Registrar {
  Registrar(const char* s, const char* n, void (*f)()) { registry().push_back({s, n, f}); }
};
inline int32_t float_key(float v) {
  int32_t i;
  std::memcpy(&i, &v, 4);
  return i >= 0 ? i : -(i & 0x7fffffff);
}
inline bool float_eq(double a, double b) {
  const float fa = (float)a, fb = (float)b;
  if (std::isnan(fa) || std::isnan(fb)) return false;
  const long long d = (long long)float_key(fa) - (long long)float_key(fb);
  return (d < 0 ? -d : d) <= 4;
}
inline void fail(const char* file, int line, const std::string& msg) {
  char buf[64];
  std::snprintf(buf, sizeof buf, ":%d: ", line);
  throw Failure{std::string(file) + buf + msg};
}
inline int run_all(const char* filter) {
  int failed = 0, ran = 0;
  for (const Case& c : registry()) {
    const std::string full = std::string(c.suite) + "." + c.name;
    if (filter && full.find(filter) == std::string::npos) continue;
    ran++;
    try {
      c.fn();
      std::printf("[  OK  ] %s\n", full.c_str());
    } catch (const Failure& f) {
      failed++;
      std::printf("[ FAIL ] %s\n    %s\n", full.c_str(), f.what.c_str());
    } catch (const std::exception& e) {
      failed++;
      std::printf("[ FAIL ] %s\n    exception: %s\n", full.c_str(), e.what());
#ifdef MINI_TEST_EXTRA_CATCH
    MINI_TEST_EXTRA_CATCH   // (a test file's own exception types, e.g. one that carries `.msg`)
#endif
    } catch (...) {
      failed++;
      std::printf("[ FAIL ] %s\n    unknown exception\n", full.c_str());
    }
  }
  std::printf("%d tests, %d failed\n", ran, failed);
  return failed ? 1 : 0;
}
}  // namespace mini

#define TEST(suite, name)                                                   \
  static void suite##_##name##_body();                                      \
  static mini::Registrar suite##_##name##_reg(#suite, #name, suite##_##name##_body); \
  static void suite##_##name##_body()

#define TEST_F(fixture, name)                                               \
  struct fixture##_##name##_T : fixture {                                   \
    void Body();                                                            \
  };                                                                        \
  static void fixture##_##name##_run() {                                    \
    fixture##_##name##_T t;                                                 \
    t.SetUp();                                                              \
    try {                                                                   \
      t.Body();                                                             \
    } catch (...) {                                                         \
      t.TearDown();                                                         \
      throw;                                                                \
    }                                                                       \
    t.TearDown();                                                           \
  }                                                                         \
  static mini::Registrar fixture##_##name##_reg(#fixture, #name, fixture##_##name##_run); \
  void fixture##_##name##_T::Body()

#define EXPECT_TRUE(c) do { if (!(c)) mini::fail(__FILE__, __LINE__, "expected true: " #c); } while (0)
#define ASSERT_TRUE(c) EXPECT_TRUE(c)
#define EXPECT_EQ(a, b) do { if (!((a) == (b))) mini::fail(__FILE__, __LINE__, "expected equal: " #a " , " #b); } while (0)
#define ASSERT_EQ(a, b) EXPECT_EQ(a, b)
#define ASSERT_FLOAT_EQ(a, b) do { if (!mini::float_eq((a), (b))) { char _m[160]; std::snprintf(_m, sizeof _m, "float mismatch: " #a " = %.9g vs " #b " = %.9g", (double)(a), (double)(b)); mini::fail(__FILE__, __LINE__, _m); } } while (0)
#define ASSERT_NEAR_REL(a, b, rel) do { const double _a = (a), _b = (b); if (!(std::fabs(_a - _b) <= (rel) * std::fabs(_b))) { char _m[160]; std::snprintf(_m, sizeof _m, "not within %g rel: %.17g vs %.17g", (double)(rel), _a, _b); mini::fail(__FILE__, __LINE__, _m); } } while (0)
#define ASSERT_THROW(stmt, ex) do { bool _t = false; try { stmt; } catch (const ex&) { _t = true; } if (!_t) mini::fail(__FILE__, __LINE__, "expected " #ex " from: " #stmt); } while (0)
