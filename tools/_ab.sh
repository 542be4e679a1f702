set -e
for s in soup mixed terrain; do timeout -k 10 120 python tools/closest_probe.py --scene $s --steps 30 | tail -1 | cut -c1-60; done
timeout -k 10 120 python tools/closest_probe.py --scene soup --width 3840 --height 2160 --steps 10 | tail -1 | cut -c1-60
timeout -k 10 120 python tools/closest_probe.py --scene terrain --width 3840 --height 2160 --steps 10 | tail -1 | cut -c1-60
timeout -k 10 60 python tools/pt_profile.py --single 16 --passes 192 | tail -1
timeout -k 10 60 python tools/pt_profile.py --single 8 --passes 64 --algorithm 1 | tail -1
