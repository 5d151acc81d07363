run() { env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --frames-in-flight $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['frames_in_flight'], d['config']['hw_queues'], d['value'], d['ms_per_step'], d['latency_ms'], d['kernel']['pipeline_ms'])"; }
F=4; echo "F=4 hwq=8 graph"; run GPU_MAX_HW_QUEUES=8
F=4; echo "F=4 hwq=8 nograph"; run GPU_MAX_HW_QUEUES=8 MCRT_GRAPH=0
F=3; echo "F=3 hwq=4 graph"; run GPU_MAX_HW_QUEUES=4
F=3; echo "F=3 hwq=8 graph"; run GPU_MAX_HW_QUEUES=8
F=4; echo "F=4 hwq=4 graph"; run GPU_MAX_HW_QUEUES=4
F=4; echo "F=4 hwq=8 graph again"; run GPU_MAX_HW_QUEUES=8
