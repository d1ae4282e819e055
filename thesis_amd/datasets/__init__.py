"""Sensor-log loaders and the synthetic `room16` workload."""
