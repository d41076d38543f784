"""CPU model of scan_screen_lean3_kernel's LDS-DMA ring (ragmeup_amd/csrc/scan_screen.hip): the kernel hands its six-tile ring over once per
PAIR of tiles behind a counted `s_waitcnt vmcnt(NIW)` + `s_barrier`, issues the pieces of tile t + 4 during tile t into the slots of tile
t - 2 without any clamp, and reads fragments up to one tile ahead of the tile it computes.  None of that is visible to a parity test until a
box is slow enough to lose the race, so the schedule is pinned here: the constants are read from the source, the VMEM stream of a wave is
replayed with in-order retirement (vmcnt(N) = everything but the N youngest operations has completed), and every hand-over is checked."""
import os
import re

import pytest

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ragmeup_amd", "csrc", "scan_screen.hip")


def kernel_text():
    s = open(SRC).read()
    a = s.index("// ---- lean form, one barrier per TWO tiles (round 4)")
    b = s.index("// ---- K-SPLIT form of the screening scan")
    return s[a:b]


def constants():
    t = kernel_text()
    nr = int(re.search(r"NR = (\d+), NDW = NWV, NIW = 24 / NWV;", t).group(1))
    ahead = int(re.search(r"\+ half \* S_CKB\) \+ (\d+)u \* S_RT \* IMGB;", t).group(1))
    assert f"tp - {ahead} * S_RT * IMGB" in t                                   # the prologue undoes the in-loop look-ahead
    pro = len(re.findall(r"issue_part\(I(\d)\{\}, b0", t))                      # tiles issued before the loop
    assert 'asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NIW) : "memory");   // tiles 0, 1, 2' in t
    assert t.count('"n"(C::NIW)') == 2                                          # prologue and pair barrier wait alike
    assert "if (gs % (S_TS / C::NIW) == 1) issue_part(std::integral_constant<int, (P + 4) % 6>{}, tp, gs / (S_TS / C::NIW));" in t
    assert "if (gs == 4 && P % 2 == 0) refresh_gthr();" in t
    assert "if (P % 2 == 0) {                              // a pair of tiles starts" in t
    s_pre = int(re.search(r"constexpr int NW = NWV, S_PRE = (\d+);", t).group(1))
    assert 'asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(S_PRE - 1));' in t
    return {"slots": nr, "ahead": ahead, "prologue_tiles": pro, "s_pre": s_pre}


def test_constants_are_the_ones_the_model_below_replays():
    c = constants()
    assert c == {"slots": 12, "ahead": 4, "prologue_tiles": 4, "s_pre": 4}
    # rmu_api.hip keeps enough zero rows behind the image for the unclamped look-ahead: (ceil(n / 32) + ahead) * 32 - n <= 31 + 32 * ahead
    api = open(os.path.join(os.path.dirname(SRC), "rmu_api.hip")).read()
    slack = int(re.search(r"static const int64_t kSlackRows = (\d+);", api).group(1))
    assert slack >= 31 + 32 * c["ahead"]


@pytest.mark.parametrize("nw", [8, 4])
@pytest.mark.parametrize("ntiles", [1, 2, 3, 5, 6, 7, 12, 13, 40])
def test_every_hand_over_of_the_ring_is_covered_by_a_counted_wait(nw, ntiles):
    c = constants()
    niw, steps = 24 // nw, 24
    ring_tiles = c["slots"] // 2
    ops = []                                     # this wave's VMEM operations in issue order: ("R",) or ("P", tile)
    done = lambda n_outstanding: ops[:max(0, len(ops) - n_outstanding)]     # in-order retirement
    landed = lambda n_outstanding: {o[1] for o in done(n_outstanding) if o[0] == "P" and done(n_outstanding).count(o) == niw}
    ops.append(("R",))
    for t in range(c["prologue_tiles"]):
        ops += [("P", t)] * niw
    assert landed(niw) >= {0, 1, 2}              # vmcnt(NIW) before the first barrier: tiles 0, 1, 2 (the fragment prefetch reaches tile 1)
    reading = set()
    for tl in range(ntiles):
        if tl % 2 == 0:                          # pair barrier
            got = landed(niw)
            need = {t for t in (tl, tl + 1, tl + 2)}     # both tiles of the pair and the first tile of the next (cross-tile prefetch)
            assert need <= got, (tl, sorted(need - got))
            reading = {tl, tl + 1}
        # tile tl: its own fragments and, from step 24 - S_PRE on, the first S_PRE fragments of tile tl + 1
        assert tl in reading and tl + 1 in landed(niw) | reading      # the prefetch target has landed (it was part of this pair's wait)
        target = tl + c["ahead"]
        # the slot the new pieces land in must not hold a tile anybody can still read: tiles of the current pair, or tile tl + 2
        # (prefetched into at the end of tile tl + 1), i.e. positions of tl & ~1, (tl & ~1) + 1, (tl & ~1) + 2
        pair0 = tl & ~1
        busy = {(pair0 + d) % ring_tiles for d in (0, 1, 2)}
        assert target % ring_tiles not in busy, (tl, target)
        # and it must be the slot of a tile that has been consumed: tile tl - 2 (previous pair), or never used (first trips)
        assert target % ring_tiles == (tl - 2) % ring_tiles
        for gs in range(steps):
            if gs % (steps // niw) == 1:
                ops.append(("P", target))
            if gs == 4 and tl % 2 == 0:
                ops.append(("R",))
        assert sum(1 for o in ops if o == ("P", target)) == niw
    # every tile that is computed was issued exactly once per piece, in order
    issued = [o[1] for o in ops if o[0] == "P"]
    assert issued == sorted(issued) and all(issued.count(t) == niw for t in range(ntiles))


def test_the_24_pieces_of_a_tile_are_shared_out_exactly_once():
    for nw in (8, 4):
        niw = 24 // nw
        ids = sorted(n * nw + w for w in range(nw) for n in range(niw))
        assert ids == list(range(24))                                         # id = 12 * half + piece: both half-k chunks, 12 one-KiB pieces each
        # a piece is 64 lanes x 16 B = 1 KiB of the chunk's swizzled image: unit p of row i lands at physical unit p ^ ((i >> 1) & 7)
        seen = set()
        for pid in range(12):
            for lane in range(64):
                f = pid * 64 + lane
                i, p = divmod(f, 24)
                seen.add((i, p ^ ((i >> 1) & 7)))
        assert len(seen) == 32 * 24                                           # 32 rows x 24 units: every 16-byte unit of the chunk exactly once
