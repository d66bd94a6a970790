"""Block-header hash of a cached `StatelessInput` — the integrity check of /root/reference/crates/host/src/bin/cli.rs:141
(`ensure!(input.block.hash_slow() == header.hash)`), host side, no GPU.

`hash_slow()` is keccak256 of the RLP list of the header's fields in consensus order (yellow paper §4.3 + the fork additions:
EIP-1559 base fee, EIP-4895 withdrawals root, EIP-4844 blob gas pair, EIP-4788 parent beacon root, EIP-7685 requests hash; a
field that is present forces every earlier optional field to be present).  The JSON spelling is the RPC one alloy's serde uses
for `Header` (`sha3Uncles`, `miner`, hex quantities).  keccak-f[1600] is the permutation `circuits/keccak_f.py` already states
for the KECCAK-F circuit's witness; the sponge here pads with 0x01 (Keccak, not SHA-3's 0x06).

Pinned in tests/test_eth_header.py by mainnet block 0 and 1 (hashes every Ethereum client agrees on) and `hashlib.sha3_256`'s
permutation (through tests/test_keccak_circuit.py).
"""
from typing import Dict, List, Optional, Sequence, Union

from .circuits.keccak_f import keccak_f

RATE = 136

Rlp = Union[bytes, Sequence["Rlp"]]


def keccak256(data: bytes) -> bytes:
    msg = bytearray(data)
    msg.append(0x01)
    msg.extend(bytes(-len(msg) % RATE))
    msg[-1] |= 0x80
    state = [0] * 25
    for off in range(0, len(msg), RATE):
        for i in range(RATE // 8):
            state[i] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], "little")
        state = keccak_f(state)
    return b"".join(int(v).to_bytes(8, "little") for v in state[:4])


def _rlp_length(n: int, short: int) -> bytes:
    if n < 56:
        return bytes([short + n])
    be = n.to_bytes((n.bit_length() + 7) // 8, "big")
    return bytes([short + 55 + len(be)]) + be


def rlp_encode(item: Rlp) -> bytes:
    if isinstance(item, (bytes, bytearray)):
        if len(item) == 1 and item[0] < 0x80:
            return bytes(item)
        return _rlp_length(len(item), 0x80) + bytes(item)
    body = b"".join(rlp_encode(x) for x in item)
    return _rlp_length(len(body), 0xC0) + body


def _hex_bytes(v: str, what: str, size: Optional[int] = None) -> bytes:
    if not isinstance(v, str) or not v.startswith("0x"):
        raise ValueError(f"header.{what}: expected a 0x-prefixed hex string, got {v!r}")
    h = v[2:]
    b = bytes.fromhex(h if len(h) % 2 == 0 else "0" + h)
    if size is not None and len(b) != size:
        raise ValueError(f"header.{what}: {len(b)} bytes, expected {size}")
    return b


def _quantity(v, what: str) -> bytes:
    """An RLP scalar: big-endian, no leading zero bytes (zero = the empty string)."""
    if isinstance(v, str):
        n = int(v, 16) if v.startswith("0x") else int(v)
    elif isinstance(v, int) and not isinstance(v, bool):
        n = v
    else:
        raise ValueError(f"header.{what}: expected a quantity, got {v!r}")
    if n < 0:
        raise ValueError(f"header.{what}: negative")
    return n.to_bytes((n.bit_length() + 7) // 8, "big")


# (JSON key, alternative key, kind); kind: fixed byte size, "q" quantity, "b" byte string
_FIELDS = [("parentHash", None, 32), ("sha3Uncles", "ommersHash", 32), ("miner", "beneficiary", 20), ("stateRoot", None, 32),
           ("transactionsRoot", None, 32), ("receiptsRoot", None, 32), ("logsBloom", None, 256), ("difficulty", None, "q"),
           ("number", None, "q"), ("gasLimit", None, "q"), ("gasUsed", None, "q"), ("timestamp", None, "q"), ("extraData", None, "b"),
           ("mixHash", None, 32), ("nonce", None, 8)]
_OPTIONAL = [("baseFeePerGas", None, "q"), ("withdrawalsRoot", None, 32), ("blobGasUsed", None, "q"), ("excessBlobGas", None, "q"),
             ("parentBeaconBlockRoot", None, 32), ("requestsHash", None, 32)]


def _get(header: Dict, key: str, alt: Optional[str]):
    if key in header and header[key] is not None:
        return header[key]
    if alt is not None and alt in header and header[alt] is not None:
        return header[alt]
    return None


def header_is_complete(header: Dict) -> bool:
    return isinstance(header, dict) and all(_get(header, k, a) is not None for k, a, _ in _FIELDS)


def header_rlp_fields(header: Dict) -> List[bytes]:
    def enc(v, key, kind):
        if kind == "q":
            return _quantity(v, key)
        if kind == "b":
            return _hex_bytes(v, key)
        return _hex_bytes(v, key, kind)

    out = []
    for key, alt, kind in _FIELDS:
        v = _get(header, key, alt)
        if v is None:
            raise ValueError(f"header.{key}: missing")
        out.append(enc(v, key, kind))
    present = [_get(header, k, a) is not None for k, a, _ in _OPTIONAL]
    last = max((i for i, p in enumerate(present) if p), default=-1)
    for i in range(last + 1):
        key, alt, kind = _OPTIONAL[i]
        if not present[i]:
            raise ValueError(f"header.{key}: missing although the later field {_OPTIONAL[last][0]} is present")
        out.append(enc(_get(header, key, alt), key, kind))
    return out


def header_hash(header: Dict) -> str:
    """`Header::hash_slow()` as 0x-hex."""
    return "0x" + keccak256(rlp_encode(header_rlp_fields(header))).hex()
