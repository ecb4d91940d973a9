# how often does a bench process run the env kernel in its fast state (3.3 ms per 50-iteration launch) / slow state (4.0 ms), and does a
# burst of host load in front of the process matter?  alternating: plain, after 8 s of 64 busy host cores
cd /tmp; R=$GRAFT_REPO_ROOT
one() { python $R/bench.py --no-cpu-baseline --no-learner --no-actor 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l)['roofline']; print('$1 avg_launch_ms %.3f frac %.3f fill %.0f GB/s sclk %s' % (r['avg_launch_ms'], r['frac'], r['write_ceiling_gbs'], (r.get('device_state_during_repeats') or [{}])[-1].get('sclk clock speed:')))
"; }
for i in 1 2 3 4 5 6 7 8; do
  one "plain      $i"
  (for k in $(seq 1 64); do (timeout 8 sh -c "while :; do :; done" &) ; done; sleep 9)
  one "after-burn $i"
done
