"""The cached input's integrity check, /root/reference/crates/host/src/bin/cli.rs:141: `input.block.hash_slow() == header.hash`.

Pins that are NOT this repository's own output: keccak256 of the empty string, the empty-trie and empty-ommers roots, and the
hashes of Ethereum mainnet blocks 0 and 1 (field values as every client serves them)."""
import json

import pytest

from zeth_amd.eth_header import header_hash, header_is_complete, header_rlp_fields, keccak256, rlp_encode

Z32 = "0x" + "00" * 32
EMPTY_TRIE = "0x56e81f171bcc55a6ff8345e692c0f86e5b48e01b996cadc001622fb5e363b421"
EMPTY_OMMERS = "0x1dcc4de8dec75d7aab85b567b6ccd41ad312451b948a7413f0a142fd40d49347"
GENESIS_HASH = "0xd4e56740f876aef8c010b86a40d5f56745a118d0906a34e69aec8c0db1cb8fa3"
BLOCK1_HASH = "0x88e96d4537bea4d9c05d12549907b32561d3bf31f45aae734cdc119f13406cb6"

GENESIS = dict(parentHash=Z32, sha3Uncles=EMPTY_OMMERS, miner="0x" + "00" * 20,
               stateRoot="0xd7f8974fb5ac78d9ac099b9ad5018bedc2ce0a72dad1827a1709da30580f0544", transactionsRoot=EMPTY_TRIE,
               receiptsRoot=EMPTY_TRIE, logsBloom="0x" + "00" * 256, difficulty="0x400000000", number="0x0", gasLimit="0x1388",
               gasUsed="0x0", timestamp="0x0", extraData="0x11bbe8db4e347b4e8c937c1c8370e4b5ed33adb3db69cbdb7a38e1e50b1b82fa",
               mixHash=Z32, nonce="0x0000000000000042")
BLOCK1 = dict(parentHash=GENESIS_HASH, sha3Uncles=EMPTY_OMMERS, miner="0x05a56e2d52c817161883f50c441c3228cfe54d9f",
              stateRoot="0xd67e4d450343046425ae4271474353857ab860dbc0a1dde64b41b5cd3a532bf3", transactionsRoot=EMPTY_TRIE,
              receiptsRoot=EMPTY_TRIE, logsBloom="0x" + "00" * 256, difficulty="0x3ff800000", number="0x1", gasLimit="0x1388",
              gasUsed="0x0", timestamp="0x55ba4224", extraData="0x476574682f76312e302e302f6c696e75782f676f312e342e32",
              mixHash="0x969b900de27b6ac6a67742365dd65f55a0526c41fd18e1b16f1a1215c2e66f59", nonce="0x539bd4979fef1ec4")


def test_keccak256_and_rlp_known_answers():
    assert keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert "0x" + keccak256(rlp_encode(b"")).hex() == EMPTY_TRIE          # the root of the empty trie = keccak(rlp(""))
    assert "0x" + keccak256(rlp_encode([])).hex() == EMPTY_OMMERS          # keccak(rlp([]))
    # RLP's own examples (yellow paper appendix B / the ethereum wiki)
    assert rlp_encode(b"dog") == b"\x83dog"
    assert rlp_encode([b"cat", b"dog"]) == b"\xc8\x83cat\x83dog"
    assert rlp_encode(b"\x0f") == b"\x0f" and rlp_encode(b"\x80") == b"\x81\x80" and rlp_encode(b"") == b"\x80"
    assert rlp_encode([[], [[]], [[], [[]]]]) == bytes.fromhex("c7c0c1c0c3c0c1c0")
    s = b"Lorem ipsum dolor sit amet, consectetur adipisicing elit"
    assert rlp_encode(s) == b"\xb8\x38" + s
    # every sponge length around the 136-byte rate goes through the same permutation hashlib's SHA-3 uses (other padding byte):
    # the multi-block absorb is pinned by the header hashes below (508- and 533-byte inputs)
    assert len({keccak256(bytes(n)) for n in (134, 135, 136, 137, 271, 272, 273)}) == 7


def test_mainnet_block_0_and_1_hash_to_their_known_hashes():
    assert header_hash(GENESIS) == GENESIS_HASH
    assert header_hash(BLOCK1) == BLOCK1_HASH
    assert header_is_complete(GENESIS) and not header_is_complete({"gasUsed": "0x1"})
    # the consensus spellings alloy also accepts, and plain integers for quantities
    alt = dict(BLOCK1)
    alt["ommersHash"], alt["beneficiary"] = alt.pop("sha3Uncles"), alt.pop("miner")
    alt["number"], alt["timestamp"] = 1, 0x55BA4224
    assert header_hash(alt) == BLOCK1_HASH
    # one flipped bit anywhere moves the hash
    bad = dict(BLOCK1, gasLimit="0x1389")
    assert header_hash(bad) != BLOCK1_HASH


def test_fork_fields_are_appended_in_order_and_may_not_skip():
    london = dict(BLOCK1, baseFeePerGas="0x3b9aca00")
    assert len(header_rlp_fields(london)) == 16 and header_hash(london) != BLOCK1_HASH
    cancun = dict(london, withdrawalsRoot=EMPTY_TRIE, blobGasUsed="0x0", excessBlobGas="0x0", parentBeaconBlockRoot=Z32)
    f = header_rlp_fields(cancun)
    assert len(f) == 20 and f[17] == b"" and f[19] == bytes(32)           # a zero quantity is the empty string, a hash keeps its zeros
    prague = dict(cancun, requestsHash="0xe3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855")
    assert len(header_rlp_fields(prague)) == 21
    with pytest.raises(ValueError, match="withdrawalsRoot: missing"):
        header_rlp_fields(dict(london, blobGasUsed="0x0", excessBlobGas="0x0"))
    with pytest.raises(ValueError, match="stateRoot: 31 bytes"):
        header_hash(dict(BLOCK1, stateRoot="0x" + "11" * 31))
    with pytest.raises(ValueError, match="nonce"):
        header_hash(dict(BLOCK1, nonce=66))


def test_cached_input_reader_checks_the_block_hash(tmp_path):
    """cli.rs:141: a cache file whose block does not hash to the name it is stored under is refused."""
    from zeth_amd.host import read_cached_input
    doc = {"block": {"header": BLOCK1, "body": {"transactions": [], "ommers": [], "withdrawals": None}},
           "witness": {"state": [], "codes": [], "keys": [], "headers": []}}
    (tmp_path / f"input_{BLOCK1_HASH}.json").write_text(json.dumps(doc))
    got = read_cached_input(str(tmp_path), BLOCK1_HASH)
    assert got.hash_checked and got.block_number == 1 and got.gas_used == 0 and got.cycles_source == "size-estimate"
    (tmp_path / f"input_{GENESIS_HASH}.json").write_text(json.dumps(doc))            # block 1 stored under block 0's name
    with pytest.raises(ValueError, match="hashes to 0x88e96d45"):
        read_cached_input(str(tmp_path), GENESIS_HASH)
    doc["block"]["header"] = dict(BLOCK1, extraData="0x00")                              # a tampered header under the right name
    (tmp_path / f"input_{BLOCK1_HASH}.json").write_text(json.dumps(doc))
    with pytest.raises(ValueError, match="not to the hash the file is named after"):
        read_cached_input(str(tmp_path), BLOCK1_HASH)
