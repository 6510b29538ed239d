// myobatch_emu.cpp — TEST TOOLING: the translation unit of tests/emu/libmyobatch_emu.so (g++ -DMYO_EMU; myochallenge_amd/build.py).
// Shared host side + the lane-serial backend; see csrc/emu_host.h.
#ifndef MYO_EMU
#error "this file is the emulation build's translation unit: compile with -DMYO_EMU"
#endif
#define MYO_BACKEND_NAME "MYO_EMU lane-serial test build"
#define MYO_BACKEND_DENSE_NEWTON 1
#define MYO_BACKEND_BATCH_FIELDS
#include "myo_host.h"
#include "emu_host.h"
