"""Host model of the scan kernel's tile order (csrc/scan_topk.hip `ItemSeq`): with or without
XCD skew, every work item must be visited by exactly one workgroup, each workgroup's items must
ascend (the kernel's strict `>` threshold relies on it for exact id-ascending ties), and with
skew s the even workgroups take (s+1)/s times the items of the odd ones."""
import re
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class ItemSeq:
    """Line-by-line model of the device struct."""

    def __init__(self, b, G, skew):
        self.base, self.j, self.skew, self.G, self.b = 0, 0, skew, G, b
        self.lim = 1 if skew == 0 else (skew if b & 1 else skew + 1)
        self.period = G if skew == 0 else skew * G + (G >> 1)

    def next(self):
        if self.skew == 0 or self.j < self.skew:
            r = self.base + self.j * self.G + self.b
        else:
            r = self.base + self.skew * self.G + (self.b >> 1)
        self.j += 1
        if self.j == self.lim:
            self.j = 0
            self.base += self.period
        return r


def _walk(G, skew, n_items):
    seen = {}
    per_wg = []
    for b in range(G):
        seq = ItemSeq(b, G, skew)
        mine = []
        # the kernel's loop: R0 holds t, R1 the next; continue while R0's item exists
        t = seq.next()
        nxt = seq.next()
        while t < n_items:
            for it in (t, nxt):
                if it < n_items:
                    assert it not in seen, (it, b, seen[it])
                    seen[it] = b
                    mine.append(it)
            t = seq.next()
            nxt = seq.next()
        per_wg.append(mine)
    return seen, per_wg


@pytest.mark.parametrize("G,skew,n_items", [
    (256, 0, 31250), (256, 4, 31250), (256, 4, 1152), (256, 4, 1153), (256, 4, 300), (256, 1, 5000),
    (256, 7, 40000), (8, 4, 100), (304, 4, 9999), (2, 3, 17), (256, 4, 0), (256, 4, 1),
])
def test_every_item_once_and_ascending(G, skew, n_items):
    seen, per_wg = _walk(G, skew, n_items)
    assert sorted(seen) == list(range(n_items))
    for mine in per_wg:
        assert mine == sorted(mine)


def test_skew_ratio():
    _, per_wg = _walk(256, 4, 256 * 4 * 50 + 128 * 50)
    even = sum(len(m) for b, m in enumerate(per_wg) if b % 2 == 0)
    odd = sum(len(m) for b, m in enumerate(per_wg) if b % 2 == 1)
    assert even * 4 == odd * 5


def test_model_matches_the_source():
    """Guard against the model drifting from the kernel: the three formulas must appear verbatim."""
    src = open(os.path.join(ROOT, "rassengine_amd", "csrc", "scan_topk.hip")).read()
    src = re.sub(r"\s+", " ", src)
    assert "lim = skew == 0 ? 1 : ((b & 1) ? skew : skew + 1);" in src
    assert "period = skew == 0 ? G : skew * G + (G >> 1);" in src
    assert "(skew == 0 || j < skew) ? base + j * G + b : base + skew * G + (b >> 1);" in src
