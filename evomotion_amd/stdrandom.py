"""The reference's replay buffers shuffle with `std::mt19937 rand_gen(seed)` + `std::shuffle` (replay_buffer.cpp:14,21,66,83).  What
that draws is fixed by the C++ standard for the engine (MT19937, 32-bit) and by libstdc++ for std::shuffle and
std::uniform_int_distribution (GCC 9 and later: two swap positions per generator call, Lemire's multiply-shift rejection for a 32-bit
engine) — restated here so that the Python buffers draw what the reference draws for the same seed.  tests/test_stdrandom.py holds it
to the container's own g++ / libstdc++ and to the draws recorded from the compiled reference (tests/golden/*_loop_golden.txt)."""


class Mt19937:
    """std::mt19937 (MT19937, 32-bit outputs), seeded like std::mt19937(seed)"""

    def __init__(self, seed=5489):
        mt = [0] * 624
        mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self._mt, self._i = mt, 624

    def __call__(self):
        if self._i >= 624:
            mt = self._mt
            for k in range(624):
                y = (mt[k] & 0x80000000) | (mt[(k + 1) % 624] & 0x7FFFFFFF)
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self._i = 0
        y = self._mt[self._i]
        self._i += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF


def _uniform_below(gen, n):
    """std::uniform_int_distribution<unsigned long>{0, n - 1}(gen) for a 32-bit engine and n <= 2^32: libstdc++'s _S_nd (Lemire)"""
    if n == 1 << 32:
        return gen()
    product = gen() * n
    low = product & 0xFFFFFFFF
    if low < n:
        threshold = ((1 << 32) - n) % n
        while low < threshold:
            product = gen() * n
            low = product & 0xFFFFFFFF
    return product >> 32


def std_shuffle(items, gen):
    """std::shuffle(items.begin(), items.end(), gen) of libstdc++ (bits/stl_algo.h), in place; returns items"""
    n = len(items)
    if n == 0:
        return items
    if 0xFFFFFFFF // n >= n:
        i = 1
        if n % 2 == 0:
            j = _uniform_below(gen, 2)
            items[i], items[j] = items[j], items[i]
            i += 1
        while i != n:
            swap_range = i + 1
            x = _uniform_below(gen, swap_range * (swap_range + 1))      # __gen_two_uniform_ints
            a, b = x // (swap_range + 1), x % (swap_range + 1)
            items[i], items[a] = items[a], items[i]
            i += 1
            items[i], items[b] = items[b], items[i]
            i += 1
        return items
    for i in range(1, n):
        j = _uniform_below(gen, i + 1)
        items[i], items[j] = items[j], items[i]
    return items
