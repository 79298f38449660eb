"""The fixed point of k_jpeg_sync (csrc/bevw_jpeg_codec.h) as a protocol, checked under random interleavings -- no GPU.

The kernel re-decodes subsequences whose entry state is not their predecessor's exit state until nothing changes; in its tail (<= 16 listed
subsequences) one wave per STRETCH walks on into the successors ("chase").  Waves of a block run concurrently and read each other's stores at
arbitrary times, so what has to hold is a property of the protocol, not of the Huffman walk: whatever the interleaving, when no wave asks
for another round every entry state is the sequential decoder's.  This model replaces the walk of subsequence j by a random function F_j of a
small state space (with the self-synchronising ones constant), yields at every memory operation, and schedules the waves at random.

Round 4's soak found the variant in which a wave that reached another wave's stretch looked at the successor's entry state before it asked
for another round (profiles/r04/README.md section 13): the model fails for it too, which is what makes the model worth keeping."""
import random

import pytest


def run(seed, N=60, S=6, tail=16, stretch_test_first=True, sync_p=0.35):
    rng = random.Random(seed)
    F = []
    for j in range(N):
        if rng.random() < sync_p:
            F.append([rng.randrange(S)] * S)          # a subsequence inside which every decoder synchronises
        else:
            F.append([rng.randrange(S) for _ in range(S)])
    seg = [j == 0 or rng.random() < 0.05 for j in range(N)]   # first subsequence of a restart segment: known entry state
    true_in = [0] * N
    for j in range(1, N):
        true_in[j] = 0 if seg[j] else F[j - 1][true_in[j - 1]]
    entry = [0] * N
    exit_ = [F[j][0] for j in range(N)]               # k_jpeg_sync0: every subsequence from the guessed state
    rounds = 0
    while True:
        listed = [j for j in range(1, N) if not seg[j] and exit_[j - 1] != entry[j]]
        if not listed:
            break
        rng.shuffle(listed)
        changed = [False]

        def lane(j):                                   # the branch for long lists: one lane per listed subsequence
            i = exit_[j - 1]; yield
            entry[j] = i; yield
            o = F[j][i]; yield
            if o != exit_[j]:
                yield
                exit_[j] = o
                changed[0] = True

        def wave(first, starts):                       # the tail: one wave per stretch
            j, i = first, exit_[first - 1]
            yield
            while True:
                o = F[j][i]; yield
                entry[j] = i; yield
                exit_[j] = o; yield
                j += 1
                if j >= N or seg[j]:
                    return
                if stretch_test_first:
                    if j in starts:
                        changed[0] = True
                        return
                    e = entry[j]; yield
                    if e == o:
                        return
                else:                                  # (the variant the soak caught)
                    e = entry[j]; yield
                    if e == o:
                        return
                    if j in starts:
                        changed[0] = True
                        return
                i = o

        if len(listed) <= tail:
            starts = set(j for j in listed if (j - 1) not in listed)
            if len(starts) < len(listed):
                changed[0] = True                      # listed, but left to the predecessor's wave
            running = [wave(j, starts) for j in starts]
        else:
            running = [lane(j) for j in listed]
        while running:
            c = rng.choice(running)
            try:
                next(c)
            except StopIteration:
                running.remove(c)
        rounds += 1
        if not changed[0]:
            break
        assert rounds <= 10 * N, "no fixed point"
    return all(seg[j] or (entry[j] == true_in[j] and exit_[j] == F[j][true_in[j]]) for j in range(1, N))


@pytest.mark.parametrize("N,S,tail,sync_p", [(60, 6, 16, 0.35), (40, 3, 16, 0.5), (120, 6, 4, 0.3), (30, 2, 16, 0.3)])
def test_every_interleaving_ends_in_the_sequential_decoders_states(N, S, tail, sync_p):
    assert all(run(seed, N, S, tail, True, sync_p) for seed in range(400))


def test_the_model_catches_the_variant_the_soak_caught():
    assert not all(run(seed, 60, 6, 16, False, 0.35) for seed in range(4000))


def _prefix_serial(part, reset):
    """What thread 0 used to do alone (k_jpeg_sync): out[i] = what thread i starts from; a thread whose range holds a reset passes on what it
    accumulated since its (last) reset."""
    out, carry = [], 0
    for t, r in zip(part, reset):
        out.append(carry)
        carry = t if r else carry + t
    return out


def _prefix_two_levels(part, reset, wave=64):
    """The same in two levels, as the kernel does it now: lane 0 of every wave walks its wave's entries from 0, thread 0 walks the waves' totals,
    and an entry behind a reset inside its own wave does not take the wave's incoming carry."""
    n = len(part)
    local, seen_before, totals, any_reset = [0] * n, [False] * n, [], []
    for w0 in range(0, n, wave):
        carry, seen = 0, False
        for i in range(w0, min(w0 + wave, n)):
            local[i], seen_before[i] = carry, seen
            carry = part[i] if reset[i] else carry + part[i]
            seen = seen or reset[i]
        totals.append(carry)
        any_reset.append(seen)
    wave_in = _prefix_serial(totals, any_reset)
    return [local[i] + (0 if seen_before[i] else wave_in[i // wave]) for i in range(n)]


def test_two_level_prefix_with_resets_equals_the_serial_walk():
    rng = random.Random(5)
    for trial in range(300):
        n = 1024
        p_reset = rng.choice([0.0, 0.002, 0.02, 0.3, 1.0])
        part = [rng.randrange(-50, 400) for _ in range(n)]
        reset = [rng.random() < p_reset for _ in range(n)]
        assert _prefix_two_levels(part, reset) == _prefix_serial(part, reset), trial
