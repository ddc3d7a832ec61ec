"""CPU-only checks of the address arithmetic of the hand-written tower convolution (sejonggo_amd/csrc/sgo_conv8w.hpp):
the LDS images filled by LDS-DMA (swizzle on the SOURCE address) are what the fragment reads expect, and the reads are
bank-conflict-free for the ds_read_b128 / ds_read_b64 lane groups of MI355X (MI355X_MICROARCH.md, LDS table).  The formulas
below restate the kernel's macros (SGW_STAGE_W / SGW_STAGE_B / SGW_READ_A / SGW_READ_B / SGW_STAGE_SKIP and the epilogue);
the GPU parity tests check the results, these check the claims the design rests on."""
import itertools

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
               [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]
LZ = 147456


def b128_cycles(addr_of_lane):
    """LDS-array cycles of one ds_read_b128 wave instruction: per lane group, the largest number of DISTINCT addresses that
    share a 16-byte slot of the 256-byte bank row (identical addresses broadcast)."""
    tot = 0
    for g in B128_GROUPS:
        slots = {}
        for l in g:
            a = addr_of_lane[l]
            slots.setdefault((a % 256) // 16, set()).add(a)
        tot += max(len(v) for v in slots.values())
    return tot


def window_read_addr(lane, wr, G, mt, ks, shift, halo):
    rl = halo + wr * 64 + (lane & 15) + shift            # rowA + shift
    c0 = (((lane >> 4) ^ rl) & 7) << 4
    return G * 16384 + (rl << 7) + (c0 ^ (64 * ks)) + mt * 2048, rl


def test_window_image_matches_fragment_reads_and_is_conflict_free():
    halo = 18                                             # w = 17
    # fill: piece id, lane -> LDS row id*8 + (lane>>3), physical chunk lane&7 holds logical chunk (lane&7) ^ ((lane>>3)&7)
    image = {}
    for idn in range(40):
        for lane in range(64):
            row = idn * 8 + (lane >> 3)
            image[(row, lane & 7)] = (row, (lane & 7) ^ ((lane >> 3) & 7))   # content: (pixel row, logical 16-B chunk)
    for wr, G, mt, ks in itertools.product((0, 1), (0, 1), range(4), (0, 1)):
        for shift in (-18, -17, -16, -1, 0, 1, 16, 17, 18):
            addrs = {}
            for lane in range(64):
                a, rl = window_read_addr(lane, wr, G, mt, ks, shift, halo)
                addrs[lane] = a
                row, pc = a // 128, (a % 128) // 16
                # MFMA fragment of lane: pixel row (lane&15) of tile mt, k elements 8*(lane>>4) + 32*ks .. +7 = logical chunk
                assert image[(row, pc)] == (rl + G * 128 + mt * 16, (lane >> 4) + 4 * ks)
            assert b128_cycles(addrs) == 4, (wr, G, mt, ks, shift)


def test_zero_redirect_keeps_the_bank_slot():
    halo = 18
    for shift in (-18, -1, 0, 17):
        for ks in (0, 1):
            addrs = {}
            for lane in range(64):
                a, rl = window_read_addr(lane, 1, 0, 2, ks, shift, halo)
                z = LZ + ((rl & 1) << 7) + (((((lane >> 4) ^ rl) & 7) << 4) ^ (64 * ks)) + 2 * 2048
                assert z % 256 == a % 256 and LZ <= z < LZ + 3 * 2048 + 256
                addrs[lane] = z if lane % 3 == 0 else a   # an arbitrary subset of lanes is off the board
            assert b128_cycles(addrs) == 4


def test_weight_image_matches_fragment_reads_and_is_conflict_free():
    image = {}
    for G, wid, i, lane in itertools.product((0, 1), range(8), (0, 1), range(64)):
        row = (wid * 2 + i) * 8 + (lane >> 3)             # row of the 128-row granule G
        c = (lane & 7) ^ ((i << 2) | (lane >> 4))          # logical chunk fetched by this lane (boff00 ^ 64 i)
        assert ((i << 2) | (lane >> 4)) == (row >> 1) & 7
        image[(G, row, lane & 7)] = (G * 128 + row, c)
    for wc, G, nt, ks in itertools.product(range(4), (0, 1), (0, 1), (0, 1)):
        addrs = {}
        for lane in range(64):
            frag = ((lane >> 4) ^ ((lane >> 1) & 7)) << 4
            a = ((wc * 32 + (lane & 15)) * 128 + frag) ^ (64 * ks)
            a += nt * 2048
            addrs[lane] = a
            assert image[(G, a // 128, (a % 128) // 16)] == (G * 128 + wc * 32 + nt * 16 + (lane & 15), (lane >> 4) + 4 * ks)
        assert b128_cycles(addrs) == 4


def test_epilogue_image_round_trip_and_b64_conflicts():
    # skip rows by DMA: instruction j of wave wid, lane -> row (wid*8+j)*2 + (lane>>5), physical chunk lane&31 holds logical
    # chunk (lane&31) ^ (row&15); the in-place pass addresses (pixel row, 4-channel piece); the copy-out reads linear 16 B
    image = {}
    for wid, j, lane in itertools.product(range(8), range(8), range(64)):
        row = (wid * 8 + j) * 2 + (lane >> 5)
        image[(row, lane & 31)] = (row, (lane & 31) ^ (row & 15))
    assert len(image) == 128 * 32
    for wr, wc, mt, qn, nt in itertools.product((0, 1), range(4), range(4), (0, 1), (0, 1)):
        addrs = []
        for lane in range(64):
            epx = (wr * 64 + (lane & 15)) * 512 + ((lane >> 4) & 1) * 8
            epc = ((wc * 4 + (lane >> 5)) ^ (lane & 15)) << 4
            a = epx + (epc ^ (nt << 5)) + mt * 8192 + qn * 256
            addrs.append(a)
            row, pc, half = a // 512, (a % 512) // 16, (a % 16) // 8
            ch = qn * 128 + wc * 32 + nt * 16 + (lane >> 4) * 4   # the lane's 4 output channels
            assert image[(row, pc)] == (wr * 64 + mt * 16 + (lane & 15), ch * 2 // 16) and half == (ch * 2 % 16) // 8
        for grp in (range(0, 32), range(32, 64)):                 # ds_read_b64: 2 x 32 lanes, 64 banks of 4 B
            banks = [(addrs[l] % 256) // 8 for l in grp]
            assert len(set(banks)) == 32
    # copy-out: wave wid, j, lane reads LDS bytes wid*8192 + j*1024 + lane*16 and stores to row / logical chunk below
    for wid, j, lane in itertools.product(range(8), range(8), range(64)):
        a = wid * 8192 + j * 1024 + lane * 16
        row = wid * 16 + j * 2 + (lane >> 5)
        assert image[(a // 512, (a % 512) // 16)] == (row, (lane & 31) ^ (j * 2 + (lane >> 5)))


def test_packed_filter_bank_feeds_the_same_fragments_as_the_lds_route():
    """k_conv4r (csrc/sgo_conv4r.hpp) loads its weight fragments from a bank in fragment order (k_prepack).  Restated here:
    piece i of the bank (16 bytes) -> (filter, tap, first input channel).  Claims: (1) the pieces of the four wave banks are a
    permutation of the filter bank's 16-byte pieces (nothing lost, nothing twice); (2) lane l of fragment (lo / hi, nt, ks) of
    K-tile t holds what the same lane reads from k_conv4w's LDS image -- filter = half * 128 + qn * 64 + wc * 32 + nt * 16 +
    (l & 15), input channels cc * 64 + ks * 32 + (l >> 4) * 8 .. + 8 of tap T -- so both kernels issue identical MFMAs;
    (3) a wave's load group (4 fragments) is 4 KB contiguous and its K-tiles follow each other in loop order."""
    WAVE_BANK = 36 * 8192
    seen = set()
    for i in range(4 * WAVE_BANK // 16):
        lane, ks, nt, qn = i & 63, (i >> 6) & 1, (i >> 7) & 1, (i >> 8) & 1
        t, g = (i >> 9) % 36, (i >> 9) // 36
        cc, T = t // 9, t % 9
        half, wc = g >> 1, g & 1
        filt = half * 128 + qn * 64 + wc * 32 + nt * 16 + (lane & 15)
        ci = cc * 64 + ks * 32 + (lane >> 4) * 8
        # (2): k_conv4w's image row / chunk for this lane (test above): granule G = qn, row wc*32 + nt*16 + (l & 15), chunk (l >> 4) + 4 ks
        assert filt == half * 128 + qn * 64 + (wc * 32 + nt * 16 + (lane & 15)) and ci == cc * 64 + ((lane >> 4) + 4 * ks) * 8
        assert 0 <= filt < 256 and 0 <= ci <= 248 and ci % 8 == 0
        seen.add((filt, T, ci))
        # (3): byte address inside the wave bank = K-tile * 8192 + group * 4096 + fragment * 1024 + lane * 16
        assert i * 16 == g * WAVE_BANK + t * 8192 + qn * 4096 + (nt * 2 + ks) * 1024 + lane * 16
    assert len(seen) == 256 * 9 * 32                     # (1)
