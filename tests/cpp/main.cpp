#include "wave_test.hpp"
int main(int argc, char **argv) {
    return ::testing::RunAll(argc, argv);
}
