"""The route consensus of select_decode's shared segments (csrc/select_decode.hpp, the G > 1 branch), model-checked on the CPU.

What thread 0 of each of the G partner workgroups does, step by step as the kernel does it (every shared-memory access is its
own atomic step, so that the partners interleave between any two of them):
    [veto]    if the slice could not be held whole:  compare-and-swap(route, 0 -> TOURNAMENT)
    [arrive]  (the slice's histogram is in the segment's)  arrived += 1
    [spin]    read arrived: >= G -> propose COOP | read route: != 0 -> propose TOURNAMENT | time out (any time) -> propose TOURNAMENT
    [decide]  prev = compare-and-swap(route, 0 -> proposal);  my route = proposal if prev == 0 else prev
Claims of the kernel's comments, checked over EVERY interleaving and every timeout choice for G = 2 and 3:
    (1) all partners end up with the same route, whatever times out;
    (2) a partner that leaves with COOP does so only after ALL histograms are in (the threshold it derives is the segment's);
    (3) a vetoing partner forces TOURNAMENT for everybody;
    (4) without timeouts and vetoes the route is COOP (the barrier is not vacuous);
    (5) nobody can be stuck: every state has a successor until all have decided (a timeout is always enabled).
This is a model of the protocol, not of the GPU: the kernel's own runs under ODTK_SELECT_COOP_TICKS=1 / 0 are
tests/test_gpu_select_routes.py."""
import itertools

COOP, TOURNAMENT = 1, 2


def explore(G, whole, allow_timeout=True):
    """All terminal outcomes {tuple(routes)} reachable; asserts the invariants on the way."""
    # per workgroup: pc in ('veto', 'arrive', 'spin_arrived', 'spin_route', 'decide', 'done'), proposal, route
    start = (tuple(('veto' if not whole[g] else 'arrive', 0, 0) for g in range(G)), 0, 0, tuple(0 for _ in range(G)))
    seen, stack, outcomes = {start}, [start], set()
    while stack:
        wgs, arrived, route, hist = stack.pop()
        succ = []
        for g, (pc, prop, mine) in enumerate(wgs):
            def with_(new, arrived=arrived, route=route, hist=hist):
                return (wgs[:g] + (new,) + wgs[g + 1:], arrived, route, hist)
            if pc == 'veto':
                succ.append(with_(('arrive', 0, 0), route=route or TOURNAMENT))
            elif pc == 'arrive':
                succ.append(with_(('spin_arrived', 0, 0), arrived=arrived + 1, hist=hist[:g] + (1,) + hist[g + 1:]))
            elif pc == 'spin_arrived':
                succ.append(with_(('decide', COOP, 0)) if arrived >= G else with_(('spin_route', 0, 0)))
                if allow_timeout:
                    succ.append(with_(('decide', TOURNAMENT, 0)))       # the clock may run out at any point of the loop
            elif pc == 'spin_route':
                succ.append(with_(('decide', TOURNAMENT, 0)) if route != 0 else with_(('spin_arrived', 0, 0)))
                if allow_timeout:
                    succ.append(with_(('decide', TOURNAMENT, 0)))
            elif pc == 'decide':
                new_route = route or prop
                mine = prop if route == 0 else route
                if mine == COOP:
                    assert all(hist) and arrived == G, 'COOP before every histogram was in'          # (2)
                succ.append(with_(('done', prop, mine), route=new_route))
        if not succ:
            assert all(w[0] == 'done' for w in wgs), 'stuck before everybody decided'                 # (5)
            outcomes.add(tuple(w[2] for w in wgs))
            continue
        for s in succ:
            if s not in seen:
                seen.add(s)
                stack.append(s)
    return outcomes


def test_partners_never_disagree():
    for G in (2, 3):
        for whole in itertools.product((True, False), repeat=G):
            outcomes = explore(G, whole)
            for routes in outcomes:
                assert len(set(routes)) == 1, (G, whole, routes)                                      # (1)
                if not all(whole):
                    assert routes[0] == TOURNAMENT, (G, whole, routes)                                # (3)
            if all(whole):
                assert {r[0] for r in outcomes} == {COOP, TOURNAMENT}                                 # both reachable with timeouts


def test_without_timeouts_the_barrier_completes():
    for G in (2, 3):
        assert explore(G, (True,) * G, allow_timeout=False) == {(COOP,) * G}                          # (4)
