"""end-to-end (PCIe-inclusive) rate of the batch API: host pools -> quicked_batch_create (H2D) -> run -> scores (D2H)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
b = datagen.generate(N, 10000, 0.05)
p = capi.make_params(algo=capi.BANDED, only_score=True)
rb = capi.ResidentBatch(b); rb.run(p, sync=True); rb.close()        # warm: pools, code objects
for label, src in (("pageable", b), ("pinned", capi.pinned_copy(b))):
  for rep in range(2):
    t0 = time.perf_counter(); rb = capi.ResidentBatch(src); t1 = time.perf_counter()
    rb.run(p, sync=True); t2 = time.perf_counter(); s, st = rb.scores(); t3 = time.perf_counter(); rb.close()
    print(f"{label}: create(H2D) {1e3*(t1-t0):.1f} ms  run {1e3*(t2-t1):.1f} ms  scores {1e3*(t3-t2):.1f} ms  -> {N/(t3-t0):,.0f} pairs/s end to end")

# packed wire formats: the client holds serialized words (serialization is not timed: it is done once,
# where the data is produced or stored); 4x / 2.7x fewer bytes over PCIe and no pack stage
for wire, name in ((capi.WIRE_2BIT, "2-bit"), (capi.WIRE_PLANES3, "planes3")):
    pw, po = capi.wire_pack_pool(b.pattern_pool, b.pattern_off, b.pattern_len, wire)
    tw, to = capi.wire_pack_pool(b.text_pool, b.text_off, b.text_len, wire)
    (pwp, h1), (twp, h2) = capi.pinned_array(pw), capi.pinned_array(tw)
    for label, (x, y) in (("pageable", (pw, tw)), ("pinned", (pwp, twp))):
      for rep in range(2):
        t0 = time.perf_counter(); rb = capi.ResidentBatch.from_wire(b, wire, x, po, y, to); t1 = time.perf_counter()
        rb.run(p, sync=True); t2 = time.perf_counter(); s2, st2 = rb.scores(); t3 = time.perf_counter(); rb.close()
        assert (s2 == s).all()
        print(f"{name} {label}: create(H2D) {1e3*(t1-t0):.1f} ms  run {1e3*(t2-t1):.1f} ms  scores {1e3*(t3-t2):.1f} ms  -> {N/(t3-t0):,.0f} pairs/s end to end")
    capi.lib().quicked_host_free(h1); capi.lib().quicked_host_free(h2)
