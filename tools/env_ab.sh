# times bench.py's default frame with an environment variable set to each given value (development aid)
# usage: tools/env_ab.sh VAR value [value ...]      (WORKLOAD selects another bench workload)
var=$1; shift
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v timeout -k 10 200 python3 bench.py --workload ${WORKLOAD:-soup1m_4k} --no-cpu-baseline --steps 3 2>/dev/null | python3 tools/bench_line.py || exit 1
done
