"""A THIRD restatement of the reference's decoder, in plain Python (test infrastructure; slow: small streams
only).  Written independently of oracle/xlz_oracle.c and kept as simple as the Go code reads, so that the
oracle's less obvious choices -- what survives a chunk that runs out of input inside a packet, the window
that Reset does not clear -- are checked by a second pair of eyes that shares no code with it.

Follows: range_decoder.go:27-131 (rc), bit_tree_decoder.go:18-70 (trees), len_decoder.go:34-60,
decompress.go:13-1136 (one packet; state and reps mutated exactly where the reference mutates them, so an
io.EOF from any ReadByte leaves what the reference leaves), reader2.go:100-250 (chunks), window.go (Window of
tests/lzma_craft.py: same uncleared Reset)."""
from lzma_craft import Window

K_TOP = 1 << 24
OK, OK_INPUT_EOF, ERR_RESULT, ERR_PROPS, ERR_HEADER_EOF, ERR_RC_INIT, ERR_UNEXPECTED_EOF = 0, 1, -1, -2, -3, -4, -5


class _Eof(Exception):
    pass


class _Rc:
    def __init__(self, data, pos, limit):
        self.d, self.p, self.lim = data, pos, limit
        self.range, self.code = 0xFFFFFFFF, 0

    def byte(self):
        if self.p >= self.lim:
            raise _Eof
        b = self.d[self.p]
        self.p += 1
        return b

    def init(self):  # range_decoder.go:27-46
        if self.byte() != 0:
            return False
        for _ in range(4):
            self.code = ((self.code << 8) | self.byte()) & 0xFFFFFFFF
        return True

    def norm(self):
        if self.range < K_TOP:
            b = self.byte()  # (may raise: the probability update in front of it has happened, :57-98)
            self.range = (self.range << 8) & 0xFFFFFFFF
            self.code = ((self.code << 8) | b) & 0xFFFFFFFF

    def bit_nn(self, probs, i):
        p = probs[i]
        bound = (self.range >> 11) * p
        if self.code < bound:
            self.range = bound
            probs[i] = p + ((2048 - p) >> 5)
            return 0
        self.code -= bound
        self.range -= bound
        probs[i] = p - (p >> 5)
        return 1

    def bit(self, probs, i):
        b = self.bit_nn(probs, i)
        self.norm()
        return b

    def direct(self, n):  # :100-131 (range / code are written back only at the end there: an EOF inside
        r = 0             # leaves them as they were -- irrelevant, the next chunk re-initialises both)
        for _ in range(n):
            self.range >>= 1
            self.code = (self.code - self.range) & 0xFFFFFFFF
            t = (0 - (self.code >> 31)) & 0xFFFFFFFF
            self.code = (self.code + (self.range & t)) & 0xFFFFFFFF
            r = ((r << 1) + ((t + 1) & 0xFFFFFFFF)) & 0xFFFFFFFF
            self.norm()
        return r


def _tree(rc, probs, base, n):
    m = 1
    for _ in range(n):
        m = (m << 1) | rc.bit(probs, base + m)
    return m - (1 << n)


def _rtree(rc, probs, base, n):
    m, sym = 1, 0
    for i in range(n):
        b = rc.bit(probs, base + m)
        m = (m << 1) | b
        sym |= b << i
    return sym


class _State:
    def __init__(self, lc, lp, pb):
        self.lc, self.lp, self.pb = lc, lp, pb
        self.reset()

    def reset(self):  # state.go:79-121
        self.lit = [1024] * (0x300 << (self.lc + self.lp))
        self.is_match, self.r0long = [1024] * 192, [1024] * 192
        self.is_rep, self.g0, self.g1, self.g2 = ([1024] * 12 for _ in range(4))
        self.slot, self.pd, self.al = [1024] * 256, [1024] * 115, [1024] * 16
        self.len = [[1024, 1024], [1024] * 128, [1024] * 128, [1024] * 256]
        self.rlen = [[1024, 1024], [1024] * 128, [1024] * 128, [1024] * 256]
        self.state = 0
        self.reps = [0, 0, 0, 0]


def _len(rc, L, ps):
    if rc.bit(L[0], 0) == 0:
        return _tree(rc, L[1], ps << 3, 3)
    if rc.bit(L[0], 1) == 0:
        return 8 + _tree(rc, L[2], ps << 3, 3)
    return 16 + _tree(rc, L[3], 0, 8)


def _run(rc, st, w, left, stop_at=None):
    """decompress.go:13 ff. until the chunk's end; 'end' / 'marker' / 'err'; raises _Eof where a ReadByte fails.
    left: bytesLeft (None = undefined).  stop_at: return 'stop' in front of the packet that starts at that output
    position (tests that compare the decoder's state in mid-stream)."""
    defined = left is not None
    while True:
        if stop_at is not None and len(w.total) >= stop_at:
            return "stop", left
        if defined and left == 0 and rc.code == 0:
            return "end", left
        ps = w.pos & ((1 << st.pb) - 1)
        s2 = (st.state << 4) + ps
        if rc.bit(st.is_match, s2) == 0:
            if defined and left == 0:
                return "err", left
            prev = 0 if w.empty() else w.get(1)
            base = 0x300 * (((w.pos & ((1 << st.lp) - 1)) << st.lc) + (prev >> (8 - st.lc)))
            sym = 1
            if st.state >= 7:
                mb = w.get(st.reps[0] + 1)
                while sym < 0x100:
                    mbit = (mb >> 7) & 1
                    mb = (mb << 1) & 0xFF
                    b = rc.bit(st.lit, base + ((1 + mbit) << 8) + sym)
                    sym = (sym << 1) | b
                    if mbit != b:
                        break
            while sym < 0x100:
                sym = (sym << 1) | rc.bit(st.lit, base + sym)
            w.put(sym - 0x100)
            st.state = 0 if st.state < 4 else (st.state - 3 if st.state < 10 else st.state - 6)
            if defined:
                left -= 1
            continue
        if rc.bit(st.is_rep, st.state) == 0:
            st.reps = [st.reps[0]] + st.reps[:3]        # :216 -- rep0 keeps its value until the distance is in
            ln = _len(rc, st.len, ps)
            st.state = 7 if st.state < 7 else 10        # :431 -- in front of the distance
            slot = _tree(rc, st.slot, min(ln, 3) << 6, 6)
            if slot < 4:
                d = slot
            else:
                nb = (slot >> 1) - 1
                d = (2 | (slot & 1)) << nb
                if slot < 14:
                    d += _rtree(rc, st.pd, d - slot, nb)
                else:
                    d += rc.direct(nb - 4) << 4
                    d += _rtree(rc, st.al, 0, 4)
            st.reps[0] = d & 0xFFFFFFFF
            if st.reps[0] == 0xFFFFFFFF:
                if rc.code == 0 and not (defined and left > 0):
                    return "marker", left
                return "err", left
            if defined and left == 0:
                return "err", left
            if st.reps[0] >= w.size or not (w.full or st.reps[0] <= w.pos):
                return "err", left
            ln += 2
        else:
            if defined and left == 0:
                return "err", left
            if w.empty():
                return "err", left
            if rc.bit(st.g0, st.state) == 0:
                if rc.bit(st.r0long, s2) == 0:
                    st.state = 9 if st.state < 7 else 11
                    w.put(w.get(st.reps[0] + 1))
                    if defined:
                        left -= 1
                    continue
            else:
                if rc.bit_nn(st.g1, st.state) == 0:
                    st.reps[0], st.reps[1] = st.reps[1], st.reps[0]   # rotated BEFORE the normalisation, :785-798
                    rc.norm()
                else:
                    rc.norm()
                    i = 2 if rc.bit_nn(st.g2, st.state) == 0 else 3
                    d = st.reps.pop(i)
                    st.reps.insert(0, d)
                    rc.norm()
            ln = _len(rc, st.rlen, ps) + 2
            st.state = 8 if st.state < 7 else 11
        trunc = False
        if defined and (left & 0xFFFFFFFF) < ln:
            ln = left & 0xFFFFFFFF
            trunc = True
        for _ in range(ln):
            w.put(w.get(st.reps[0] + 1))
        if defined:
            left -= ln
        if trunc:
            return "err", left


def lzma2_raw(data, dict_size):
    """-> (output bytes, status, input consumed); no output cap."""
    w = Window(dict_size)
    st = None
    pos, h5 = 0, 0
    n = len(data)

    def done(status, p):
        return bytes(w.total), status, p
    while True:
        if pos >= n:
            return done(ERR_UNEXPECTED_EOF, pos)
        c = data[pos]
        pos += 1
        if c == 0 or 3 <= c < 0x80:
            return done(OK, pos)
        hl = 3 if c < 3 else (6 if (c >> 5) >= 6 else 5)
        if n - pos < hl - 1:
            return done(ERR_UNEXPECTED_EOF, n)
        unc = (data[pos] << 8) | data[pos + 1]
        if c == 1 or (c >> 5) == 7:
            w.reset()
        if c < 3:
            pos += 2
            k = min(unc + 1, n - pos)
            for b in data[pos:pos + k]:
                w.put(b)
            pos += k
            continue
        unc = (((c & 0x1F) << 16) | unc) + 1
        comp = ((data[pos + 2] << 8) | data[pos + 3]) + 1
        pos += 4
        sub = c >> 5
        if sub >= 6:
            h5 = data[pos]
            pos += 1
        first = st is None
        if first or sub >= 6:
            if h5 >= 225:
                return done(ERR_PROPS, pos)
            st = _State(h5 % 9, (h5 // 9) % 5, h5 // 45)
        elif sub == 5:
            st.reset()
        rc = _Rc(data, pos, min(pos + comp, n))
        try:
            if not rc.init():
                return done(ERR_RC_INIT, rc.p)
        except _Eof:
            return done(ERR_HEADER_EOF if first else OK_INPUT_EOF, rc.p)
        try:
            res, _ = _run(rc, st, w, unc)
        except _Eof:
            res = "eof"
        pos = rc.p
        if res == "err":
            return done(ERR_RESULT, pos)


def lzma1_alone(data):
    """NewReader1 + io.Copy (reader1.go:18-61,149-159,223-254) -> (output bytes, status, input consumed)"""
    n = len(data)
    if n < 13:
        return b"", ERR_HEADER_EOF, n
    if data[0] >= 225:
        return b"", ERR_PROPS, 13
    ds = int.from_bytes(data[1:5], "little")
    size = int.from_bytes(data[5:13], "little")
    w = Window(max(ds, 4096))                        # reader1.go:199-201
    st = _State(data[0] % 9, (data[0] // 9) % 5, data[0] // 45)
    rc = _Rc(data, 13, n)
    try:
        if not rc.init():
            return b"", ERR_RC_INIT, rc.p
    except _Eof:
        return b"", ERR_HEADER_EOF, rc.p
    try:
        res, _ = _run(rc, st, w, None if size == (1 << 64) - 1 else size)
    except _Eof:
        return bytes(w.total), OK_INPUT_EOF, rc.p
    return bytes(w.total), (ERR_RESULT if res == "err" else OK), rc.p
