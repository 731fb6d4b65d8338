// One translation unit per shared AMOEBA test body: compiled with -DTEST_HEADER='"TestAmoebaVdwForce.h"' etc.  The test body is
// the reference's own file, included from $(REF)/plugins/amoeba/tests at build time (never copied) -- see tests/hip/Makefile.
#include "HipAmoebaTests.h"
#include TEST_HEADER

void runPlatformTests() {
}
