// Minimal test harness for the C++ mirror tests (no framework in the image).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

struct TestCase { const char* name; std::function<void()> fn; };
inline std::vector<TestCase>& registry() { static std::vector<TestCase> r; return r; }
struct Registrar { Registrar(const char* n, std::function<void()> f) { registry().push_back({n, std::move(f)}); } };
#define TEST(name) static void name(); static Registrar reg_##name(#name, name); static void name()
#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "  CHECK failed: %s  (%s:%d)\n", #cond, __FILE__, __LINE__); throw 1; } } while (0)
#define CHECK_EQ(a, b) do { auto va_ = (a); auto vb_ = (b); if (!(va_ == vb_)) { std::fprintf(stderr, "  CHECK_EQ failed: %s == %s  (%s:%d)\n", #a, #b, __FILE__, __LINE__); throw 1; } } while (0)
// the call must throw ibu::IbuError of the given kind; binds it to `e` for payload checks in `body`
#define CHECK_THROWS(kind_, expr, body) do { bool thrown_ = false; try { expr; } catch (const ibu::IbuError& e) { thrown_ = true; \
  if (e.kind() != ibu::IbuError::kind_) { std::fprintf(stderr, "  wrong error kind %s, wanted %s (%s:%d)\n", e.name(), #kind_, __FILE__, __LINE__); throw 1; } body; } \
  if (!thrown_) { std::fprintf(stderr, "  expected IbuError::%s from %s (%s:%d)\n", #kind_, #expr, __FILE__, __LINE__); throw 1; } } while (0)

inline int run_all(int argc, char** argv) {
  int failed = 0, ran = 0;
  for (auto& t : registry()) {
    if (argc > 1 && std::string(t.name).find(argv[1]) == std::string::npos) continue;
    ++ran;
    try { t.fn(); std::printf("ok   %s\n", t.name); }
    catch (const std::exception& e) { ++failed; std::printf("FAIL %s: %s\n", t.name, e.what()); }
    catch (...) { ++failed; std::printf("FAIL %s\n", t.name); }
  }
  std::printf("%d run, %d failed\n", ran, failed);
  return failed ? 1 : 0;
}
