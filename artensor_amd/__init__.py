"""artensor_amd: MI355X-native numerical contraction engine behind artensor's executor API."""
