#!/bin/bash
# engine-path ms of the fitted models against the planner's price of a pivot (AASR_PG_PIVOT_COST / _COST3: rows a pivot
# of the first / second part must rescue)
for kind in stationary speechlike; do
  for cost in 200 400 615 900 1300 2000; do
    for c3 in 30 100; do
    AASR_PG_PIVOT_COST=$cost AASR_PG_PIVOT_COST3=$c3 timeout 300 python tools/bench_fitted.py $kind 5 2>/dev/null | grep "parts\|engine path" | cut -c1-400 | tr '\n' ' '; echo " [cost $cost / $c3 $kind]"
    done
  done
done
