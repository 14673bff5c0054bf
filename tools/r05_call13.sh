R=$GRAFT_REPO_ROOT; cd $R
for lib in "" "$R/build_variants/lib_noom.so" "" "$R/build_variants/lib_noom.so"; do MFX_LIB_PATH=$lib timeout 300 python bench.py --dtype fp16x2 --legs none --no-cpu-baseline --no-families 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('fp16x2 lib=[$lib]',d['value'],d['ms_per_step'])"; done
for lib in "" "$R/build_variants/lib_noom.so"; do MFX_LIB_PATH=$lib timeout 300 python bench.py --legs none --no-cpu-baseline --no-families 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('bf16 lib=[$lib]',d['value'],d['ms_per_step'])"; done
