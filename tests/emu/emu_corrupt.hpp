// Damaged input for the emulated decoders (test infrastructure): when EMU_CORRUPT=<seed> is set, the drivers flip / overwrite bytes of
// what the oracle encoded before the kernels read it and skip their comparisons -- what is checked then is that the kernels neither
// crash (the test builds these runs with -fsanitize=address: LDS arrays are globals, device buffers heap blocks), nor deadlock (the
// emulator's scheduler notices), nor run away.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <vector>

static inline bool emu_corrupt_on() { return getenv("EMU_CORRUPT") != nullptr; }

static inline void emu_corrupt(uint8_t* p, size_t n, unsigned salt)
{
    if (!emu_corrupt_on() || n == 0) return;
    uint32_t x = (uint32_t)atoi(getenv("EMU_CORRUPT")) * 2654435761u + salt * 40503u + 12345u;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x >> 8; };
    switch (rnd() % 4) {
    case 0: for (int k = 0; k < 1 + (int)(rnd() % 4); k++) p[rnd() % n] ^= (uint8_t)(1u << (rnd() % 8)); break;          // bit flips
    case 1: for (int k = 0; k < 1 + (int)(rnd() % 8); k++) p[rnd() % n] = (uint8_t)rnd(); break;                          // bytes
    case 2: { const size_t a = rnd() % n, len = 1 + rnd() % 64; for (size_t i = a; i < n && i < a + len; i++) p[i] = (uint8_t)rnd(); } break;   // a range
    default: { const size_t a = rnd() % n; for (size_t i = a; i < n; i++) p[i] = 0; } break;                              // cut off
    }
}
