"""An executable MODEL of the lazily chained launches' protocol, run under random interleavings (round-5 verdict, weak point 13: "≈ 460 lines
of host logic guarding a device-side protocol ... no model or invariant checker exists beyond the tests").  No GPU, no product code: the rules
below restate, with their places in the product,

  * the host side of a chained launch — ring slot, lap of the ring, predecessor link, "does this launch bring a kernel of its own",
    publish kernel on the control stream — rayaccel_amd/csrc/racc_launch.inc `launchTraverse` (the `chain` / `lazy` branches),
  * the publish kernel — racc_hip.hip `chainPublishKernel` (fields, own `next` = 0, then the predecessor's `next`, in that order),
  * a wave of a chained kernel — racc_kernel_v8.inc, the refill block: look-before-take on a wave's first draw, chunks by fetch-add on the
    batch's cursor word, "exhausted", the walk down `next` only for a wave that has loaded rays, the two tag bits (four batches side by side in a
    wave: a tag is handed out again only when no ray in flight carries it), exit when exhausted with nothing in flight and no successor,
  * completion — racc_launch.inc `finishChainLocked`: publish kernels and chain kernels ended, a batch is complete iff its cursor has passed its
    count, the first incomplete one gets a catch-up kernel that starts on its descriptor and carries on down the links; repeat,

and a scheduler draws, step by step, which of the enabled actions happens next: the host issues the next batch or waits, a publish kernel runs
(control stream: in order), a kernel at the head of a lane's stream starts, a wave refills, a ray in flight ends.  Checked at every step and at
every return of a wait: a ray index is handed out at most once, a descriptor is only read through a link written in this lap of the ring, a
result lands in the array of the batch its ray belongs to (the tag-to-array map of the wave that traced it), no kernel outlives the lap in which
it was launched; after a wait EVERY ray of every batch issued before it has been traced exactly once.  What the model cannot see: memory
ordering below "a link is read after the fields it covers" (the product's release store / relaxed loads), and the kernels' arithmetic."""
import random

import pytest

RING = 8            # (the product's ring has 256 slots; a small one laps within a test)
AUTO_LANES = 3      # racc_hip.hip: autoLanes


class Kernel:
    def __init__(self, kid, slot, count, chunk, waves, lap, chain_id):
        self.kid, self.lap, self.chain_id = kid, lap, chain_id
        self.started = False
        self.waves = [Wave(self, slot, count, chunk) for _ in range(waves)]

    @property
    def ended(self):
        return self.started and all(w.gone for w in self.waves)


class Wave:
    """racc_kernel_v8.inc: one wave's view of the chain.  Its own batch comes from the kernel's ARGUMENTS (count, cursor word, results), only
    `next` from its descriptor — the kernel may start before its publish kernel has run."""
    def __init__(self, kernel, slot, count, chunk):
        self.k = kernel
        self.cur, self.count, self.chunk = slot, count, chunk     # chainCur, count; the cursor word is cursors[self.cur]
        self.res = {0: slot}                                      # tag -> the batch whose results array the tag selects (res0..res3)
        self.tag = 0
        self.exhausted = self.ever_loaded = self.gone = False
        self.beg = self.end = 0                                   # the chunk being handed to lanes
        self.inflight = []                                        # (slot, index, tag)


class Model:
    def __init__(self, rng, batches, max_count, waves, lanes):
        self.rng = rng
        self.plan = [(rng.randint(1, max_count), rng.choice([1, 2, 3, 5])) for _ in range(batches)]      # (rays, chunk) per batch
        self.waves_per_kernel, self.lanes = waves, lanes
        self.ring = [dict(count=0, next=0, fields=False) for _ in range(RING)]      # ChainDesc: fields valid?, next = 1 + slot
        self.cursors = [0] * RING
        self.head = 0                     # chainHead
        self.lap = 0
        self.last = None                  # chainLast: slot of the launch issued before, while nobody has waited for it
        self.chain_id = 0
        self.live = []                    # chainLive: kernels launched and not seen ended
        self.outstanding = []             # chainOutstanding: slots published since the last finishChain
        self.host_desc = {}               # chainHost[slot]
        self.publish_q = []               # control stream, in order: (slot, count, pred, lap)
        self.streams = [[] for _ in range(lanes + 1)]      # per lane (+ the catch-up lane): kernels in stream order
        self.next_lane = 0
        self.kid = 0
        self.traced = {}                  # (launch number, index) -> times handed out
        self.slot_launch = {}             # slot -> launch number of the batch living there in this lap
        self.issued = 0
        self.waiting = False
        self.steps = 0
        self.slow_control_stream = rng.random() < 0.4      # some schedules let the publish kernels lag far behind the traversal kernels

    # ------------------------------------------------------------------------------------------------ host: racc_launch.inc launchTraverse
    def host_issue(self):
        count, chunk = self.plan[self.issued]
        slot = self.head % RING
        if slot == 0 and self.head != 0:
            # a lap of the ring: everything outstanding is completed, every stream drained, cursors and descriptors zeroed (launchTraverse)
            if not self.finish_chain_step_until_done():
                return False              # (the wait inside the lap needs the GPU to make progress first)
            assert all(k.ended for st in self.streams for k in st), "lap of the ring with a kernel still running"
            self.streams = [[] for _ in self.streams]
            self.cursors = [0] * RING
            self.ring = [dict(count=0, next=0, fields=False) for _ in range(RING)]
            self.last = None
            self.lap += 1
            self.slot_launch = {}
        pred = self.last if self.last is not None else -1
        if pred < 0:
            self.chain_id += 1
        self.live = [k for k in self.live if not k.ended]                      # hipEventQuery on the kernels' end events
        own = pred < 0 or sum(1 for k in self.live if k.chain_id == self.chain_id) < AUTO_LANES
        self.slot_launch[slot] = self.issued
        if own:
            k = Kernel(self.kid, slot, count, chunk, self.waves_per_kernel, self.lap, self.chain_id)
            self.kid += 1
            self.streams[self.next_lane % AUTO_LANES].append(k)
            self.live.append(k)
        self.next_lane += 1
        self.publish_q.append((slot, count, pred, self.lap))
        self.host_desc[slot] = (count, chunk)
        self.outstanding.append(slot)
        self.last = slot
        self.head += 1
        self.issued += 1
        return True

    # ------------------------------------------------------------------------------------------------ host: finishChainLocked, one round
    def finish_chain_step_until_done(self):
        """One attempt of finishChainLocked's loop body: True when everything outstanding is complete.  (The product BLOCKS in the stream
        synchronisations; the model returns False and is called again after the GPU has moved.)"""
        if self.publish_q or any(not k.ended for k in self.live):
            return False
        self.live = []
        for slot in self.outstanding:
            if self.cursors[slot] < self.host_desc[slot][0]:
                count, chunk = self.host_desc[slot]
                k = Kernel(self.kid, slot, count, chunk, self.waves_per_kernel, self.lap, self.chain_id)      # launchCatchUp: starts on that descriptor
                self.kid += 1
                self.streams[-1].append(k)
                self.live.append(k)
                return False
        self.outstanding = []
        return True

    def host_wait_done(self):
        # racc_hip_wait(LANE_AUTO) has returned: every batch issued so far is traced, exactly once
        for n in range(self.issued):
            count = self.plan[n][0]
            for i in range(count):
                assert self.traced.get((n, i), 0) == 1, "after a wait: ray %d of batch %d traced %d times" % (i, n, self.traced.get((n, i), 0))
        assert not any(w.inflight for st in self.streams for k in st for w in k.waves)
        self.last = None                  # (launchPending cleared: the next launch starts a chain of its own)

    # ------------------------------------------------------------------------------------------------ GPU: chainPublishKernel
    def run_publish(self):
        slot, count, pred, lap = self.publish_q.pop(0)
        assert lap == self.lap, "a publish kernel of the lap before ran after the ring was zeroed"
        d = self.ring[slot]
        d["count"], d["fields"], d["next"] = count, True, 0
        if pred >= 0:
            self.ring[pred]["next"] = slot + 1

    # ------------------------------------------------------------------------------------------------ GPU: a wave (racc_kernel_v8.inc, refill)
    def wave_refill(self, w):
        assert w.k.lap == self.lap, "a wave of the lap before is still running after the ring was zeroed"
        if w.beg == w.end:
            for tries in range(2):
                if not w.exhausted:
                    if not w.ever_loaded and self.cursors[w.cur] >= w.count:          # look before taking (a wave's first draw)
                        w.exhausted = True
                        w.beg = w.end = w.count
                        break
                    b = self.cursors[w.cur]
                    self.cursors[w.cur] += w.chunk                                      # atomic fetch-add
                    w.exhausted = b >= w.count
                    w.beg = w.count if w.exhausted else b
                    w.end = w.count if w.exhausted else min(b + w.chunk, w.count)
                if not w.exhausted or tries:
                    break
                if not w.ever_loaded:
                    break
                nx = self.ring[w.cur]["next"]
                if not nx:
                    break
                new_tag = (w.tag + 1) & 3
                if any(t == new_tag for _, _, t in w.inflight):
                    break                                                               # a ray of the batch four back is still in flight
                d = self.ring[nx - 1]
                assert d["fields"], "a link leads to a descriptor whose fields were not written in this lap"
                w.cur, w.count = nx - 1, d["count"]
                w.chunk = self.host_desc[nx - 1][1]                                     # (the product keeps a.chunk; any chunk is correct)
                w.res[new_tag] = nx - 1
                w.tag = new_tag
                w.exhausted = False
        take = min(self.rng.randint(1, 3), w.end - w.beg)                              # as many idle lanes as the step happens to have
        for i in range(w.beg, w.beg + take):
            key = (self.slot_launch[w.cur], i)
            self.traced[key] = self.traced.get(key, 0) + 1
            assert self.traced[key] == 1, "ray %d of batch %d handed out twice" % (i, key[0])
            w.inflight.append((w.cur, i, w.tag))
            w.ever_loaded = True
        w.beg += take
        if w.exhausted and w.beg == w.end and not w.inflight:
            w.gone = True

    def ray_ends(self, w):
        slot, _, tag = w.inflight.pop(self.rng.randrange(len(w.inflight)))
        assert w.res[tag] == slot, "a result went to the array of batch slot %d, its ray belongs to slot %d" % (w.res[tag], slot)

    # ------------------------------------------------------------------------------------------------ the scheduler
    def enabled(self):
        acts = []
        if self.waiting:
            acts.append(("wait",))
        else:
            if self.issued < len(self.plan):
                acts.append(("issue",))
            if self.issued and (self.outstanding or self.last is not None):
                acts.append(("begin_wait",))
        n_before = len(acts)
        for st in self.streams:
            for k in st:
                if not k.ended:
                    if not k.started:
                        acts.append(("start", k))
                    else:
                        for w in k.waves:
                            if not w.gone:
                                acts.append(("refill", w))
                                if w.inflight:
                                    acts.append(("ray", w))
                    break                 # stream order: the kernels behind it wait
        if self.publish_q and (not self.slow_control_stream or len(acts) == n_before or self.rng.random() < 0.03):
            acts.append(("publish",))
        return acts

    def run(self, max_steps=200000):
        while True:
            self.steps += 1
            assert self.steps < max_steps, "no termination"
            acts = self.enabled()
            gpu = [a for a in acts if a[0] in ("publish", "start", "refill", "ray")]
            if not self.waiting and self.issued == len(self.plan) and not self.outstanding and self.last is None and not gpu:
                return
            # the host is usually far ahead of the GPU, sometimes behind it: both regimes
            a = self.rng.choice(acts if self.rng.random() < 0.5 or not gpu else gpu)
            if a[0] == "issue":
                self.host_issue()
            elif a[0] == "begin_wait":
                self.waiting = True
            elif a[0] == "wait":
                if self.finish_chain_step_until_done():
                    self.host_wait_done()
                    self.waiting = False
            elif a[0] == "publish":
                self.run_publish()
            elif a[0] == "start":
                a[1].started = True
            elif a[0] == "refill":
                self.wave_refill(a[1])
            elif a[0] == "ray":
                self.ray_ends(a[1])


@pytest.mark.parametrize("seed", range(40))
def test_random_interleavings_of_the_lazy_chain(seed):
    """25 schedules per seed x 40 seeds: up to 30 batches of 1 .. 40 rays in chunks of 1 .. 5, kernels of 1 .. 4 waves, a ring of 8 slots (up to
    three laps), waits wherever the scheduler puts them."""
    for sub in range(25):
        rng = random.Random(seed * 1000 + sub)
        m = Model(rng, batches=rng.randint(1, 30), max_count=rng.choice([3, 12, 40]), waves=rng.randint(1, 4), lanes=4)
        m.run()
        assert m.issued == len(m.plan) and not m.outstanding
        m.host_wait_done()


def test_the_model_catches_a_broken_protocol():
    """The checker is not vacuous: without the tag rule (a tag handed out again while a ray of the batch four back is in flight) results land in
    the wrong batch's array, and without the catch-up kernel a batch the chain did not reach stays untraced — the model says so."""
    class NoTagRule(Model):
        def wave_refill(self, w):
            saved = w.inflight
            w.inflight = []               # the wave does not look at what is in flight
            try:
                Model.wave_refill(self, w)
            finally:
                w.inflight = saved + w.inflight
            w.gone = False if w.inflight else w.gone

    class NoCatchUp(Model):
        def finish_chain_step_until_done(self):
            if self.publish_q or any(not k.ended for k in self.live):
                return False
            self.live, self.outstanding = [], []
            return True

    for broken, message in ((NoTagRule, "its ray belongs to"), (NoCatchUp, "after a wait")):
        caught = 0
        for seed in range(300):
            rng = random.Random(seed)
            m = broken(rng, batches=rng.randint(8, 30), max_count=40, waves=rng.randint(1, 3), lanes=4)
            try:
                m.run()
                m.host_wait_done()
            except AssertionError as e:
                assert message in str(e), str(e)
                caught += 1
        assert caught > 0, broken.__name__
