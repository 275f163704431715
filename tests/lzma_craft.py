"""Test infrastructure: a minimal LZMA packet ENCODER, used to craft streams that no real
compressor writes (a match as the very first packet, rep matches that reach behind an LZMA2
dictionary reset, distances at the edge of the dictionary ...).

It is the mirror image of the decoder the reference implements (decompress.go:8-1136): the same
probability tables and contexts, a range ENCODER instead of the decoder, and the caller says which
packets to emit.  The window is the reference's (window.go): a circular buffer that Reset does not
clear, so that the literal contexts of crafted streams are the ones the reference will see.
Nothing here is used by the product; expected outputs of the tests come from the oracle.
"""

K_TOP = 1 << 24


class RangeEncoder:
    def __init__(self):
        self.low = 0
        self.range = 0xFFFFFFFF
        self.cache = 0
        self.cache_size = 1
        self.out = bytearray()

    def _shift_low(self):
        if self.low < 0xFF000000 or self.low >= (1 << 32):
            carry = self.low >> 32
            temp = self.cache
            while True:
                self.out.append((temp + carry) & 0xFF)
                temp = 0xFF
                self.cache_size -= 1
                if self.cache_size == 0:
                    break
            self.cache = (self.low >> 24) & 0xFF
        self.cache_size += 1
        self.low = (self.low & 0x00FFFFFF) << 8

    def bit(self, probs, idx, bit):
        p = probs[idx]
        bound = (self.range >> 11) * p
        if bit == 0:
            self.range = bound
            probs[idx] = p + ((2048 - p) >> 5)
        else:
            self.low += bound
            self.range -= bound
            probs[idx] = p - (p >> 5)
        while self.range < K_TOP:
            self.range = (self.range << 8) & 0xFFFFFFFF
            self._shift_low()

    def direct(self, value, nbits):
        for i in range(nbits - 1, -1, -1):
            self.range >>= 1
            if (value >> i) & 1:
                self.low += self.range
            while self.range < K_TOP:
                self.range = (self.range << 8) & 0xFFFFFFFF
                self._shift_low()

    def finish(self):
        for _ in range(5):
            self._shift_low()
        return bytes(self.out)


class Window:
    """window.go:8-159"""

    def __init__(self, size):
        self.buf = bytearray(size)
        self.size = size
        self.pos = 0
        self.full = False
        self.total = bytearray()

    def put(self, b):
        self.buf[self.pos] = b
        self.pos += 1
        if self.pos >= self.size:
            self.pos -= self.size
            self.full = True
        self.total.append(b)

    def get(self, dist):
        i = self.pos - dist if dist <= self.pos else self.size - dist + self.pos
        return self.buf[i]

    def empty(self):
        return self.pos == 0 and not self.full

    def reset(self):
        self.pos = 0
        self.full = False


def _bittree(rc, probs, base, nbits, value):
    m = 1
    for i in range(nbits - 1, -1, -1):
        b = (value >> i) & 1
        rc.bit(probs, base + m, b)
        m = (m << 1) | b


def _bittree_rev(rc, probs, base, nbits, value):
    m = 1
    for i in range(nbits):
        b = (value >> i) & 1
        rc.bit(probs, base + m, b)
        m = (m << 1) | b


class LenCoder:
    def __init__(self):
        self.choice = [1024, 1024]
        self.low = [1024] * (16 << 3)
        self.mid = [1024] * (16 << 3)
        self.high = [1024] * 256

    def encode(self, rc, length, pos_state):  # length already minus kMatchMinLen
        if length < 8:
            rc.bit(self.choice, 0, 0)
            _bittree(rc, self.low, pos_state << 3, 3, length)
        elif length < 16:
            rc.bit(self.choice, 0, 1)
            rc.bit(self.choice, 1, 0)
            _bittree(rc, self.mid, pos_state << 3, 3, length - 8)
        else:
            rc.bit(self.choice, 0, 1)
            rc.bit(self.choice, 1, 1)
            _bittree(rc, self.high, 0, 8, length - 16)


class Encoder:
    """Packets in, range-coded LZMA payload out.  `window` may be shared between the chunks of an
    LZMA2 stream; `new_chunk()` starts a fresh range coder while the model lives on."""

    def __init__(self, lc=3, lp=0, pb=2, dict_size=1 << 16, window=None):
        self.w = window or Window(dict_size)
        self.rc = RangeEncoder()
        self.renew(lc, lp, pb)

    def renew(self, lc, lp, pb):  # state.Renew / newState + Reset (state.go:47-121)
        self.lc, self.lp, self.pb = lc, lp, pb
        self.reset_state()

    def reset_state(self):
        self.lit = [1024] * (0x300 << (self.lc + self.lp))
        self.is_match = [1024] * (12 << 4)
        self.is_rep = [1024] * 12
        self.is_rep_g0 = [1024] * 12
        self.is_rep_g1 = [1024] * 12
        self.is_rep_g2 = [1024] * 12
        self.is_rep0_long = [1024] * (12 << 4)
        self.pos_slot = [1024] * (4 << 6)
        self.pos_dec = [1024] * 115
        self.align = [1024] * 16
        self.len = LenCoder()
        self.rep_len = LenCoder()
        self.state = 0
        self.reps = [0, 0, 0, 0]

    def new_chunk(self):
        self.rc = RangeEncoder()

    def _ctx(self):
        pos_state = self.w.pos & ((1 << self.pb) - 1)
        return pos_state, (self.state << 4) + pos_state

    def literal(self, byte):
        pos_state, s2 = self._ctx()
        self.rc.bit(self.is_match, s2, 0)
        prev = 0 if self.w.empty() else self.w.get(1)
        lit_state = ((self.w.pos & ((1 << self.lp) - 1)) << self.lc) + (prev >> (8 - self.lc))
        base = 0x300 * lit_state
        symbol = 1
        if self.state >= 7:
            mb = self.w.get(self.reps[0] + 1)
            i = 7
            while i >= 0:
                match_bit = (mb >> i) & 1
                b = (byte >> i) & 1
                self.rc.bit(self.lit, base + ((1 + match_bit) << 8) + symbol, b)
                symbol = (symbol << 1) | b
                i -= 1
                if match_bit != b:
                    break
            while i >= 0:
                b = (byte >> i) & 1
                self.rc.bit(self.lit, base + symbol, b)
                symbol = (symbol << 1) | b
                i -= 1
        else:
            for i in range(7, -1, -1):
                b = (byte >> i) & 1
                self.rc.bit(self.lit, base + symbol, b)
                symbol = (symbol << 1) | b
        self.w.put(byte)
        self.state = 0 if self.state < 4 else (self.state - 3 if self.state < 10 else self.state - 6)

    def _copy(self, dist, length):
        for _ in range(length):
            self.w.put(self.w.get(dist))

    def _distance(self, dist, len_state):  # dist = rep0 value (distance - 1)
        if dist < 4:
            slot = dist
        else:
            n = dist.bit_length()
            slot = ((n - 1) << 1) | ((dist >> (n - 2)) & 1)
        _bittree(self.rc, self.pos_slot, len_state << 6, 6, slot)
        if slot >= 4:
            nbits = (slot >> 1) - 1
            base = (2 | (slot & 1)) << nbits
            if slot < 14:
                _bittree_rev(self.rc, self.pos_dec, base - slot, nbits, dist - base)
            else:
                self.rc.direct((dist - base) >> 4, nbits - 4)
                _bittree_rev(self.rc, self.align, 0, 4, (dist - base) & 15)

    def match(self, distance, length, copy=True):
        """simple match: distance >= 1, 2 <= length <= 273"""
        pos_state, s2 = self._ctx()
        self.rc.bit(self.is_match, s2, 1)
        self.rc.bit(self.is_rep, self.state, 0)
        self.reps = [distance - 1] + self.reps[:3]
        self.len.encode(self.rc, length - 2, pos_state)
        self.state = 7 if self.state < 7 else 10
        self._distance(distance - 1, min(length - 2, 3))
        if copy:
            self._copy(distance, length)

    def end_marker(self):
        pos_state, s2 = self._ctx()
        self.rc.bit(self.is_match, s2, 1)
        self.rc.bit(self.is_rep, self.state, 0)
        self.len.encode(self.rc, 0, pos_state)
        self.state = 7 if self.state < 7 else 10
        self._distance(0xFFFFFFFF, 0)

    def short_rep(self):
        pos_state, s2 = self._ctx()
        self.rc.bit(self.is_match, s2, 1)
        self.rc.bit(self.is_rep, self.state, 1)
        self.rc.bit(self.is_rep_g0, self.state, 0)
        self.rc.bit(self.is_rep0_long, s2, 0)
        self.state = 9 if self.state < 7 else 11
        self._copy(self.reps[0] + 1, 1)

    def rep(self, idx, length):
        """rep match with reps[idx], 2 <= length <= 273"""
        pos_state, s2 = self._ctx()
        self.rc.bit(self.is_match, s2, 1)
        self.rc.bit(self.is_rep, self.state, 1)
        if idx == 0:
            self.rc.bit(self.is_rep_g0, self.state, 0)
            self.rc.bit(self.is_rep0_long, s2, 1)
        else:
            self.rc.bit(self.is_rep_g0, self.state, 1)
            if idx == 1:
                self.rc.bit(self.is_rep_g1, self.state, 0)
            else:
                self.rc.bit(self.is_rep_g1, self.state, 1)
                self.rc.bit(self.is_rep_g2, self.state, idx - 2)
            d = self.reps.pop(idx)
            self.reps.insert(0, d)
        self.rep_len.encode(self.rc, length - 2, pos_state)
        self.state = 8 if self.state < 7 else 11
        self._copy(self.reps[0] + 1, length)

    def payload(self):
        return self.rc.finish()


def props_byte(lc, lp, pb):
    return (pb * 5 + lp) * 9 + lc


def alone_header(lc, lp, pb, dict_size, size=None):
    return bytes([props_byte(lc, lp, pb)]) + dict_size.to_bytes(4, "little") + \
        (b"\xff" * 8 if size is None else size.to_bytes(8, "little"))


# ---- LZMA2 framing (reader2.go:100-214) -----------------------------------------------------
def lzma2_stored(data, dict_reset):
    assert 1 <= len(data) <= 65536
    return bytes([1 if dict_reset else 2]) + (len(data) - 1).to_bytes(2, "big") + bytes(data)


def lzma2_lzma_chunk(control, unc_size, payload, props=None):
    """control: 0x80 nothing reset, 0xA0 state reset, 0xC0 + new props, 0xE0 + dictionary reset"""
    assert 1 <= unc_size <= (1 << 21) and 1 <= len(payload) <= 65536
    u = unc_size - 1
    h = bytes([control | (u >> 16), (u >> 8) & 0xFF, u & 0xFF]) + (len(payload) - 1).to_bytes(2, "big")
    if control >= 0xC0:
        h += bytes([props])
    return h + payload


# ---- random LZMA2 streams with every kind of chunk and reset (differential fuzzing) --------------
SMALL_PROPS = [(3, 0, 2), (0, 0, 0), (1, 1, 1), (0, 2, 0), (4, 0, 0)]
# what reader2.go:159-165 accepts beyond the LZMA2 format's lc+lp <= 4: any lc <= 8, lp <= 4 (models up to 0x300 << 12)
ANY_PROPS = SMALL_PROPS + [(4, 3, 1), (8, 0, 0), (5, 1, 3), (8, 4, 2), (6, 4, 0), (3, 4, 4)]


def random_lzma2_stream(rnd, dict_size=4096, max_chunks=8, max_packets=120, props=SMALL_PROPS):
    """A structurally valid LZMA2 stream made of random packets, cut into chunks with random control
    bytes: stored chunks with and without dictionary reset, LZMA chunks that reset nothing / the
    state / state + properties / everything.  Rep matches are free to reach behind a dictionary
    reset (the bytes the reference's uncleared window still holds there).  -> (bytes, expected output
    by the crafter's own window model).  props: the (lc, lp, pb) sets chunks choose from."""
    lc, lp, pb = rnd.choice(props)
    w = Window(dict_size)
    e = None
    out = bytearray()
    first_lzma = True
    for c in range(rnd.randint(1, max_chunks)):
        kind = rnd.random()
        if kind < 0.3:  # stored chunk
            data = bytes(rnd.randrange(256) for _ in range(rnd.choice([1, 2, 7, 60, 300, 2000])))
            reset = rnd.random() < 0.5 or c == 0
            if reset:
                w.reset()
            for b in data:
                w.put(b)
            out += lzma2_stored(data, reset)
            continue
        if first_lzma:
            control = 0xE0 if rnd.random() < 0.8 else 0xC0
        else:
            control = rnd.choice([0x80, 0x80, 0x80, 0xA0, 0xC0, 0xE0])
        if control >= 0xC0:
            lc, lp, pb = rnd.choice(props)
        if control == 0xE0:
            w.reset()
        if e is None:
            e = Encoder(lc, lp, pb, dict_size, window=w)
        else:
            e.new_chunk()
            if control >= 0xC0:
                e.renew(lc, lp, pb)
            elif control == 0xA0:
                e.reset_state()
        first_lzma = False
        start = len(w.total)
        n = rnd.randint(1, max_packets)
        for _ in range(n):
            r = rnd.random()
            fill = w.size if w.full else w.pos
            if w.empty() or r < 0.4:
                e.literal(rnd.randrange(256) if rnd.random() < 0.3 else rnd.choice(b"abcde "))
            elif r < 0.6 and fill >= 1:
                d = rnd.randint(1, min(fill, dict_size))
                e.match(d, rnd.choice([2, 3, 4, 8, 9, 17, 18, 40, 64, 65, 273]))
            elif r < 0.75:
                e.short_rep()
            else:
                e.rep(rnd.randrange(4), rnd.choice([2, 3, 9, 16, 17, 70, 273]))
            if len(w.total) - start > 60_000:
                break
        pay = e.payload()
        if len(pay) > 65536 or len(w.total) == start:
            break
        out += lzma2_lzma_chunk(control, len(w.total) - start, pay, props_byte(lc, lp, pb))
    out += b"\x00"
    return bytes(out), bytes(w.total)


def long_stale_lzma2_stream(dict_size, first_epoch=2_600_000, seed=1):
    """An LZMA2 stream whose FIRST dictionary epoch is megabytes long (cheap to write: a few thousand literals, then
    maximum-length matches), followed by stored chunks that reset the dictionary and 0x80 chunks whose rep matches, short
    reps and matched literals read far behind the reset -- window bytes of an epoch that a pull reader's sliding output
    window has long dropped (window.go:135-140 keeps them in the reference's buffer).  -> (bytes, expected output)"""
    import random
    rnd = random.Random(seed)
    w = Window(dict_size)
    e = Encoder(3, 0, 2, dict_size, window=w)
    blob = b""
    control = 0xE0
    start = 0

    def flush():
        nonlocal blob, control, start
        blob += lzma2_lzma_chunk(control, len(w.total) - start, e.payload(), props_byte(3, 0, 2))
        control = 0x80
        start = len(w.total)
        e.new_chunk()
    for i in range(3000):
        e.literal(rnd.randrange(97, 123))
    while len(w.total) < first_epoch:
        fill = min(len(w.total), dict_size)
        e.match(rnd.choice([1, 3, 7, 200, 2999, fill - 1, fill // 2, fill]), 273)
        if rnd.random() < 0.05:
            e.literal(rnd.randrange(256))
        if len(w.total) - start > 900_000:
            flush()
    full = min(len(w.total), dict_size)
    e.match(full - 1, 5)      # four distinct reps for the chunks behind the reset
    e.match(77, 4)
    e.match(full // 2, 6)
    e.match(1500, 3)
    flush()
    for rounds in range(3):
        w.reset()
        data = bytes(rnd.randrange(65, 91) for _ in range(rnd.choice([2, 5, 40])))
        for b in data:
            w.put(b)
        blob += lzma2_stored(data, dict_reset=True)
        start = len(w.total)
        e.literal(0x41)           # a matched literal whose matchByte lies behind the reset
        e.rep(0, 8)
        e.literal(0x42)
        e.rep(2, 30)
        e.short_rep()
        e.rep(3, 273)
        e.rep(1, 2)
        e.match(3, 4)
        e.rep(1, 100)
        for _ in range(40):       # a longer epoch every other round: what the NEXT round reads is then a mix of two epochs
            e.match(rnd.choice([1, 2, 5]), rnd.choice([20, 273]))
            if rounds != 1:
                break
        flush()
    return blob + b"\x00", bytes(w.total)


def long_lzma1_stream(lc, lp, pb, dict_size=1 << 16, total=1_500_000, seed=1):
    """A VALID .lzma stream of about `total` bytes for ANY lc <= 8, lp <= 4 (liblzma refuses to encode lc+lp > 4): a few
    thousand literals, then long matches, rep matches and short reps with a literal here and there -- megabytes of output
    from a few thousand packets.  -> (bytes with the 13-byte header and the end marker, expected output)"""
    import random
    rnd = random.Random(seed)
    e = Encoder(lc, lp, pb, dict_size)
    for i in range(2500):
        e.literal(rnd.randrange(97, 123) if rnd.random() < 0.8 else rnd.randrange(256))
    while len(e.w.total) < total:
        fill = min(len(e.w.total), dict_size)
        r = rnd.random()
        if r < 0.55:
            e.match(rnd.choice([1, 2, 9, 333, 2400, fill - 1, fill // 3, fill]), rnd.choice([273, 273, 100, 18]))
        elif r < 0.7:
            e.rep(rnd.randrange(4), rnd.choice([273, 64, 2]))
        elif r < 0.8:
            e.short_rep()
        else:
            e.literal(rnd.randrange(256))
    want = bytes(e.w.total)
    e.end_marker()
    return alone_header(lc, lp, pb, dict_size) + e.payload(), want
