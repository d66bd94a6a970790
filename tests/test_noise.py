"""The blinding-row generator (csrc/noise.h on the device, oracle/noise.h on the CPU): ChaCha12 keyed by 256 bits, cell = six stream
words folded mod P the way upstream's `Elem::random` folds six `next_u32()` draws.  Pinned from OUTSIDE the repository: both block
functions reproduce RFC 8439 section 2.3.2's ChaCha20 test vector when asked for 10 double rounds; the fold is re-derived here in
plain Python integers.  (GPU twin of these checks: tests/test_witness_gpu.py.)"""
import numpy as np

import zko

P = 2013265921
RFC_KEY = np.frombuffer(bytes(range(32)), dtype="<u4").copy()
RFC_TAIL = np.array([1, 0x09000000, 0x4a000000, 0x00000000], dtype=np.uint32)           # counter 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00
RFC_OUT = np.array([0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3, 0xc7f4d1c7, 0x0368c033, 0x9aaa2204, 0x4e6cd4c3,
                    0x466482d2, 0x09aa9f07, 0x05d7c214, 0xa2028bd9, 0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2], dtype=np.uint32)


def chacha_py(key, tail, double_rounds):
    """RFC 8439 section 2.3 restated on Python integers (a third statement, independent of both C twins)"""
    st = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + [int(k) for k in key] + [int(t) for t in tail]
    w = list(st)
    rot = lambda v, c: ((v << c) | (v >> (32 - c))) & 0xFFFFFFFF

    def qr(a, b, c, d):
        w[a] = (w[a] + w[b]) & 0xFFFFFFFF; w[d] = rot(w[d] ^ w[a], 16)
        w[c] = (w[c] + w[d]) & 0xFFFFFFFF; w[b] = rot(w[b] ^ w[c], 12)
        w[a] = (w[a] + w[b]) & 0xFFFFFFFF; w[d] = rot(w[d] ^ w[a], 8)
        w[c] = (w[c] + w[d]) & 0xFFFFFFFF; w[b] = rot(w[b] ^ w[c], 7)
    for _ in range(double_rounds):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & 0xFFFFFFFF for a, b in zip(w, st)]


def test_python_statement_reproduces_the_rfc_vector():
    assert chacha_py(RFC_KEY, RFC_TAIL, 10) == [int(x) for x in RFC_OUT]


def test_oracle_block_function_is_rfc_8439(oracle):
    out = np.zeros(16, dtype=np.uint32)
    oracle.zko_chacha_block(RFC_KEY, RFC_TAIL, 10, out)
    assert np.array_equal(out, RFC_OUT)
    rng = np.random.default_rng(1)
    for rounds in (4, 6, 10):
        key, tail = rng.integers(0, 2**32, 8, dtype=np.uint64).astype(np.uint32), rng.integers(0, 2**32, 4, dtype=np.uint64).astype(np.uint32)
        oracle.zko_chacha_block(key, tail, rounds, out)
        assert [int(x) for x in out] == chacha_py(key, tail, rounds)


def test_noise_cell_is_six_chacha12_words_folded_mod_p(oracle):
    rng = np.random.default_rng(2)
    for _ in range(50):
        key = rng.integers(0, 2**32, 8, dtype=np.uint64).astype(np.uint32)
        group, col, row = int(rng.integers(0, 3)), int(rng.integers(0, 4000)), int(rng.integers(0, 1 << 24))
        blk = chacha_py(key, [row, col, group, 0x314e4b5a], 6)
        v = 0
        for w in blk[:6]:
            v = ((v << 32) + w) % P
        assert int.from_bytes(b"ZKN1", "little") == 0x314e4b5a
        assert oracle.zko_fp_decode(oracle.zko_noise_cell(key, group, col, row)) == v
    # distinct cells of one key differ; the same cell under another key differs
    k1, k2 = zko.key_words(0x2E80), zko.key_words(0x2E81)
    cells = {oracle.zko_noise_cell(k1, 1, c, r) for c in range(8) for r in range(64)}
    assert len(cells) == 512 and oracle.zko_noise_cell(k1, 1, 0, 0) != oracle.zko_noise_cell(k2, 1, 0, 0)


def test_noise_rows_of_a_witness_are_the_keyed_stream_and_nothing_else_moves(oracle):
    from zeth_amd.circuits import syn_air
    oc = zko.OracleCircuit(oracle, syn_air.syn_tiny())
    po2, zk = 9, 100
    code, data, out = oc.witgen(po2, zk, seed=7, noise_seed=0x2E80)
    code2, data2, out2 = oc.witgen(po2, zk, seed=7, noise_seed=(1 << 200) + 5)
    n, A = 1 << po2, (1 << po2) - zk
    wd = data.size // n
    d, d2 = data.reshape(wd, n), data2.reshape(wd, n)
    assert np.array_equal(code, code2) and np.array_equal(out, out2) and np.array_equal(d[:, :A], d2[:, :A])
    assert not np.array_equal(d[:, A:], d2[:, A:])
    key = zko.key_words(0x2E80)
    for c in (0, wd - 1):
        for r in (A, n - 1):
            assert int(d[c, r]) == oracle.zko_noise_cell(key, 2, c, r)          # group 2 = data
    assert int(data.max()) < P


def test_key_conventions_of_the_package():
    from zeth_amd import hal
    assert hal.noise_key(0) is None and hal.noise_key(None) is None and hal.noise_key(np.zeros(8, np.uint32)) is None
    assert np.array_equal(hal.noise_key(0x2E80), zko.key_words(0x2E80))
    big = int.from_bytes(bytes(range(1, 33)), "little")
    assert np.array_equal(hal.noise_key(big), np.frombuffer(bytes(range(1, 33)), dtype="<u4")) and np.array_equal(hal.noise_key(bytes(range(1, 33))), hal.noise_key(big))
    from zeth_amd.prover import fresh_noise_seed
    a, b = fresh_noise_seed(), fresh_noise_seed()
    assert a != b and a.bit_length() > 200 and hal.noise_key(a) is not None


def test_the_products_generator_equals_the_oracles_and_the_rfc(oracle):
    """csrc/noise.h compiled for the host (zkh_noise_cell_host / zkh_chacha_block_host: the same inline functions the kernels call)"""
    import ctypes as C
    from zeth_amd import hal
    lib = hal.load_library()
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
    out = np.zeros(16, dtype=np.uint32)
    lib.zkh_chacha_block_host(p(RFC_KEY), p(RFC_TAIL), 10, p(out))
    assert np.array_equal(out, RFC_OUT)
    rng = np.random.default_rng(3)
    for _ in range(200):
        key = rng.integers(0, 2**32, 8, dtype=np.uint64).astype(np.uint32)
        group, col, row = int(rng.integers(0, 3)), int(rng.integers(0, 4000)), int(rng.integers(0, 1 << 26))
        assert lib.zkh_noise_cell_host(p(key), group, col, row) == oracle.zko_noise_cell(key, group, col, row)
