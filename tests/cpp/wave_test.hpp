// Minimal gtest-compatible harness (googletest is an empty submodule in the reference
// tree and is not installed here): TEST / TEST_F / EXPECT_* / ASSERT_* and a main()
// that runs every registered case, so the reference's test bodies can be re-expressed
// almost verbatim.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

namespace testing {
class Test {
 public:
    virtual ~Test() {}
    virtual void SetUp() {}
    virtual void TearDown() {}
    virtual void TestBody() = 0;
};
struct Registry {
    struct Case {
        std::string name;
        std::function<Test *()> make;
    };
    static std::vector<Case> &cases() {
        static std::vector<Case> c;
        return c;
    }
    static int &failures() {
        static int f = 0;
        return f;
    }
};
struct Registrar {
    Registrar(const char *suite, const char *name, std::function<Test *()> make) {
        Registry::cases().push_back({std::string(suite) + "." + name, std::move(make)});
    }
};
inline int RunAll(int argc, char **argv) {
    const char *filter = argc > 1 ? argv[1] : nullptr;
    int ran = 0, failed = 0;
    for (auto &c : Registry::cases()) {
        if (filter && c.name.find(filter) == std::string::npos) continue;
        std::printf("[ RUN      ] %s\n", c.name.c_str());
        std::fflush(stdout);
        const int before = Registry::failures();
        Test *t = c.make();
        t->SetUp();
        t->TestBody();
        t->TearDown();
        delete t;
        ++ran;
        if (Registry::failures() != before) {
            ++failed;
            std::printf("[  FAILED  ] %s\n", c.name.c_str());
        } else {
            std::printf("[       OK ] %s\n", c.name.c_str());
        }
    }
    std::printf("[==========] %d tests ran, %d failed\n", ran, failed);
    return failed ? 1 : 0;
}
}  // namespace testing

#define WT_FAIL_(msg)                                                           \
    do {                                                                        \
        std::printf("%s:%d: Failure: %s\n", __FILE__, __LINE__, msg);           \
        ++::testing::Registry::failures();                                      \
    } while (0)
#define EXPECT_TRUE(c) do { if (!(c)) WT_FAIL_("expected true: " #c); } while (0)
#define EXPECT_FALSE(c) do { if (c) WT_FAIL_("expected false: " #c); } while (0)
#define EXPECT_LT(a, b) do { if (!((a) < (b))) { std::printf("  %g vs %g\n", (double) (a), (double) (b)); WT_FAIL_(#a " < " #b); } } while (0)
#define EXPECT_GT(a, b) do { if (!((a) > (b))) { std::printf("  %g vs %g\n", (double) (a), (double) (b)); WT_FAIL_(#a " > " #b); } } while (0)
#define EXPECT_EQ(a, b) do { if (!((a) == (b))) WT_FAIL_(#a " == " #b); } while (0)
#define ASSERT_TRUE(c) do { if (!(c)) { WT_FAIL_("assert: " #c); return; } } while (0)

#define TEST(suite, name)                                                               \
    class suite##_##name##_Test : public ::testing::Test {                              \
        void TestBody() override;                                                       \
    };                                                                                  \
    static ::testing::Registrar suite##_##name##_reg(#suite, #name,                     \
                                                     [] { return new suite##_##name##_Test; }); \
    void suite##_##name##_Test::TestBody()

#define TEST_F(fixture, name)                                                           \
    class fixture##_##name##_Test : public fixture {                                    \
        void TestBody() override;                                                       \
    };                                                                                  \
    static ::testing::Registrar fixture##_##name##_reg(#fixture, #name,                 \
                                                       [] { return new fixture##_##name##_Test; }); \
    void fixture##_##name##_Test::TestBody()

// test data locations (repo-relative by default; the pytest wrapper sets the cwd)
inline std::string wave_test_path(const char *rel) {
    const char *root = std::getenv("WAVE_TEST_ROOT");
    return std::string(root ? root : ".") + "/" + rel;
}
