// Identity of the source tree this library was built from: the first 16 hex digits of a SHA-256 over csrc/*.hip and
// csrc/*.hpp (names and contents, sorted by name; this file excluded), passed in by the build (ist-gcn_amd/_lib.py:
// -DISTGCN_BUILD_ID).  tools/profile_summarise.py writes the same hash into the counter summaries under profiles/, and
// bench.py quotes a summary's `traffic` / `mfma_util` only when it matches the library that is running.
#ifndef ISTGCN_BUILD_ID
#define ISTGCN_BUILD_ID "unknown"
#endif
extern "C" const char* istgcn_build_id() { return ISTGCN_BUILD_ID; }
