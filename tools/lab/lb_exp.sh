for d in uniform sorted 28bit mult256 lowbyte const; do timeout 100 python tools/one_call_time.py 1e8 3 $d 2>&1 | grep "one_read=on" | cut -c1-200; done
timeout 100 python tools/one_call_time.py 1e8 2 sorted u64 2>&1 | grep "one_read=on" | cut -c1-200
