"""Discrete-event model of the scheduler of asm_fused_kernel (csrc/asm_propagate.hip, the opt-in XCD-resident dataflow
propagator): per-XCD ticket streams, in-order field claims, completion counters, ticket prefetch and the exit rule, stepped
under random interleavings with an arbitrary number of blocks per XCD (including XCDs without any).  Test infrastructure:
tests/test_asm_dataflow_model.py holds the protocol to "every task runs exactly once, a field never leaves its XCD, every
block leaves" — the properties the kernel's correctness argument rests on."""
import random
import sys

NONE = 0x7fffffff


def simulate(F, D, n1, n2, n3, blocks_per_xcc, nslot, seed, prefetch=True):
    rng = random.Random(seed)
    nb, nc = n2 * D, n3 * D
    NT = n1 + nb + nc
    spx = min(nslot, F)
    next_field = [0]
    ticket = [0] * 8
    end1 = [0] * 8
    claim = [[0] * F for _ in range(8)]
    done = {k: [[0] * F for _ in range(8)] for k in "abc"}
    executed = {}
    # block state: (xcc, state, data)
    blocks = []
    for x, n in enumerate(blocks_per_xcc):
        for _ in range(n):
            blocks.append(dict(x=x, st="start", cur=None, nxt=None, alive=True, work=0))

    def pop(x):
        t = ticket[x]
        ticket[x] += 1
        return t

    def decode(t):
        it, r = divmod(t, NT)
        if r < n1:
            return 0, it, r
        if r < n1 + nb:
            return 1, it - 1, r - n1
        return 2, it - 2, r - n1 - nb

    steps = 0
    while any(b["alive"] for b in blocks):
        steps += 1
        if steps > 5_000_000:
            return "LIVELOCK/DEADLOCK", executed
        b = rng.choice([b for b in blocks if b["alive"]])
        x = b["x"]
        st = b["st"]
        if st == "start":
            b["cur"] = pop(x)
            b["st"] = "sched"
        elif st == "sched":
            if prefetch and b["nxt"] is None:
                b["nxt"] = pop(x)
            phase, j, tile = decode(b["cur"])
            if j < 0:
                b["st"] = "next"
                continue
            code = NONE
            if j < F:
                if phase == 0 and tile == 0:
                    prev = 1
                    if j > 0:
                        c = claim[x][j - 1]
                        if c != 0:
                            prev = c
                        elif end1[x] != 0 and j - 1 >= end1[x] - 1:
                            prev = NONE
                        else:
                            continue  # spin
                    if prev != NONE:
                        got = next_field[0]
                        next_field[0] += 1
                        code = got + 1 if got < F else NONE
                        if code == NONE:
                            end1[x] = j + 1
                    claim[x][j] = code
                else:
                    c = claim[x][j]
                    if c != 0:
                        code = c
                    elif end1[x] != 0 and j >= end1[x] - 1:
                        code = NONE
                    else:
                        continue  # spin
            if code == NONE:
                if phase == 2:
                    b["alive"] = False
                else:
                    b["st"] = "next"
                continue
            b["task"] = (phase, j, tile, code - 1)
            b["st"] = "dep"
        elif st == "dep":
            phase, j, tile, f = b["task"]
            if phase == 0:
                if j >= spx:
                    ok = done["b"][x][j - spx] >= nb if D > 1 else done["c"][x][j - spx] >= nc
                else:
                    ok = True
            elif phase == 1:
                ok = done["a"][x][j] >= n1
            else:
                ok = done["b"][x][j] >= nb
            if not ok:
                continue
            b["work"] = rng.randint(1, 4)
            b["st"] = "run"
        elif st == "run":
            b["work"] -= 1
            if b["work"] <= 0:
                phase, j, tile, f = b["task"]
                key = (f, phase, tile)
                assert key not in executed, key
                executed[key] = x
                done["abc"[phase]][x][j] += 1
                b["st"] = "next"
        elif st == "next":
            if prefetch:
                b["cur"], b["nxt"] = b["nxt"], None
            else:
                b["cur"] = pop(x)
            b["st"] = "sched"
    want = {(f, 0, t) for f in range(F) for t in range(n1)} | {(f, 1, t) for f in range(F) for t in range(nb)} | \
           {(f, 2, t) for f in range(F) for t in range(nc)}
    if set(executed) != want:
        return f"INCOMPLETE {len(want) - len(executed)} missing", executed
    # a field's tasks all ran on one XCD
    for f in range(F):
        xs = {x for (ff, _, _), x in executed.items() if ff == f}
        if len(xs) != 1:
            return f"field {f} ran on XCDs {xs}", executed
    return "ok", executed


if __name__ == "__main__":
    bad = 0
    cases = 0
    for seed in range(60):
        rng = random.Random(1000 + seed)
        F = rng.choice([1, 2, 3, 5, 8, 9, 20])
        D = rng.choice([1, 1, 2])
        n1 = rng.choice([1, 2, 4])
        n2 = rng.choice([1, 2, 4])
        bp = [rng.choice([0, 1, 2, 5]) for _ in range(8)]
        if sum(bp) == 0:
            bp[3] = 1
        for nslot in (1, 3, 4):
            r, _ = simulate(F, D, n1, n2, n1, bp, nslot, seed)
            cases += 1
            if r != "ok":
                bad += 1
                print("FAIL", dict(F=F, D=D, n1=n1, n2=n2, bp=bp, nslot=nslot, seed=seed), r)
    print(f"{cases} cases, {bad} failures")
