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


# ---- the L2 form's row norms (round 5): a second, tiny ring beside the image ring ----------------------------------------------------------------
def l2_constants():
    t = kernel_text()
    assert "if (L2N && gs >= 13 && gs <= 16) frag_wait_nrm(fr[gs % S_PRE]);" in t
    assert 'asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(S_PRE - 1 + 4));' in t
    assert "if (L2N && gs == 12) read_nrm(std::integral_constant<int, (P + 1) % 6>{});" in t
    assert "if (w == 0) issue_nrm(std::integral_constant<int, (P + 4) % 6>{}, np);" in t and "if (L2N && gs == 4 && P % 2 == 0) {" in t
    assert "issue_nrm(I0{}, np);" in t and "issue_nrm(I2{}, np + 2 * S_RT);" in t and "np += 4 * S_RT;" in t and "np += 2 * S_RT;" in t
    assert "if (L2N && (NWV == 8 || wave_live)) read_nrm(I0{});   // (in front of the fragment prefetch" in t
    return {"read_step": 12, "wait_steps": (13, 16), "dma_step": 4}


@pytest.mark.parametrize("nw", [8, 4])
@pytest.mark.parametrize("ntiles", [1, 2, 3, 6, 7, 13, 40])
def test_l2_norm_ring_lands_before_it_is_read_and_is_not_overwritten_while_readable(nw, ntiles):
    """Wave 0's VMEM stream with the norm DMAs in it (one per PAIR of tiles, two pairs ahead, next to the threshold refresh): at the pair
    barrier's vmcnt(NIW) the norms of this pair AND the next have landed (the first tile of the next pair is read for during this pair's
    second tile), and the slot a new DMA lands in holds the norms of the pair that has just been left."""
    c, l2 = constants(), l2_constants()
    niw, steps = 24 // nw, 24
    ops = [("R",), ("N", 0), ("N", 2)]                       # prologue: refresh, norms of tiles 0-1 and 2-3, then the image pieces of tiles 0..3
    for t in range(c["prologue_tiles"]):
        ops += [("P", t)] * niw
    done = lambda n_out: ops[:max(0, len(ops) - n_out)]
    norms_landed = lambda n_out: {o[1] + d for o in done(n_out) if o[0] == "N" for d in (0, 1)}
    assert {0, 1, 2, 3} <= norms_landed(niw)                 # before the first barrier: z of tile 0 is read right behind it
    read_at = {}                                             # tile -> tile during which its z is read (step 12), -1 = prologue
    read_at[0] = -1
    for tl in range(ntiles):
        if tl % 2 == 0:
            got = norms_landed(niw)
            assert {tl, tl + 1, tl + 2, tl + 3} <= got, (tl, sorted(got))
        for gs in range(steps):
            if gs % (steps // niw) == 1:
                ops.append(("P", tl + c["ahead"]))
            if gs == l2["dma_step"] and tl % 2 == 0:
                ops.append(("R",))
                pair = tl + 4                                # norms of tiles tl + 4, tl + 5 -> ring positions of tiles tl - 2, tl - 1
                assert (pair % 6, (pair + 1) % 6) == ((tl - 2) % 6, (tl - 1) % 6)
                # every read of those positions' previous contents (z of tile tl - 2 and of tile tl - 1) happened before this pair's barrier
                assert all(read_at.get(t, -2) < tl for t in (tl - 2, tl - 1) if t >= 0)
                ops.append(("N", pair))
            if gs == l2["read_step"]:
                read_at[tl + 1] = tl                         # z of the NEXT tile
                # it landed at or before this pair's barrier (tl + 1 belongs to this pair or is the first tile of the next one)
                barrier_tile = tl & ~1
                assert tl + 1 <= barrier_tile + 3


def test_l2_norm_reads_are_counted_into_the_fragment_waits():
    """LDS operations retire in order: with the four norm reads issued behind fragment read F(16) at step 12, the waits of steps 13..16 must
    leave S_PRE - 1 + 4 operations outstanding to cover exactly their own fragment, and step 17's ordinary wait also covers the norms."""
    c, l2 = constants(), l2_constants()
    s_pre = c["s_pre"]
    q = [("F", m) for m in range(s_pre)]                     # outstanding LDS reads in issue order (fragments 0..3 prefetched)
    for gs in range(24):
        allow = s_pre - 1 + (4 if l2["wait_steps"][0] <= gs <= l2["wait_steps"][1] else 0)
        while len(q) > allow:
            q.pop(0)
        assert ("F", gs) not in q, gs                        # this step's fragment has landed
        if gs == 17:
            assert not any(o[0] == "N" for o in q)           # the norms are complete well before the next tile's first MFMA
        q.append(("F", gs + s_pre))
        if gs == l2["read_step"]:
            q += [("N", i) for i in range(4)]
        # never more than the counter can hold (lgkmcnt is 4 bits)
        assert len(q) <= 15
