"""Host-side helpers of the reference's model/layers/utils.py that remain on the host:
the key -> channel-slice converter (utils.py:22-37).  NMS / top-K / POI gather (utils.py:45-145)
run on device in libmonoflex_hip.so (mfx_decode_topk / mfx_decode_boxes)."""


class Converter_key2channel(object):
    def __init__(self, keys, channels):
        self.keys = [key for key_group in keys for key in key_group]
        self.channels = [channel for channel_groups in channels for channel in channel_groups]

    def __call__(self, key):
        index = self.keys.index(key)
        s = sum(self.channels[:index])
        return slice(s, s + self.channels[index], 1)
