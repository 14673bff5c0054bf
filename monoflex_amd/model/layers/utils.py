"""Host-side helpers of the reference's model/layers/utils.py that remain on the host:
the key -> channel-slice converter (utils.py:22-37).  NMS / top-K / POI gather (utils.py:45-145)
run on device in libmonoflex_hip.so (mfx_decode_topk / mfx_decode_boxes)."""


class Converter_key2channel:
    """Channel range of a named regression head inside the concatenated `reg` map (reference API: model/layers/utils.py:22-37 -- constructed from
    cfg.MODEL.HEAD.REGRESSION_HEADS / REGRESSION_CHANNELS, both lists of groups; `converter(name)` -> slice)."""

    def __init__(self, keys, channels):
        names = [n for group in keys for n in group]
        widths = [w for group in channels for w in group]
        if len(names) != len(widths):
            raise ValueError("REGRESSION_HEADS and REGRESSION_CHANNELS disagree: %d names, %d widths" % (len(names), len(widths)))
        self._range = {}
        start = 0
        for n, w in zip(names, widths):
            self._range.setdefault(n, (start, start + w))     # (a repeated name keeps its first range, like list.index)
            start += w
        self.keys, self.channels = names, widths              # the reference's public attributes

    def __call__(self, key):
        if key not in self._range:
            raise ValueError("%r is not in list" % (key,))     # what list.index raises in the reference
        lo, hi = self._range[key]
        return slice(lo, hi, 1)
