// One translation unit per shared test body: compiled with -DTEST_HEADER='"TestNonbondedForce.h"' etc.
// The test body itself is the reference's own file, included from $(REF)/tests at build time
// (never copied into this repository) -- see tests/hip/Makefile.
#include "HipTests.h"
#include TEST_HEADER

void runPlatformTests() {
}
