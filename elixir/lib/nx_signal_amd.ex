defmodule NxSignalAMD do
  @moduledoc """
  Drop-in for the hot path of `NxSignal` (v0.3.0) on AMD Instinct MI355X: `stft/3`, `istft/3`, `as_windowed/2`,
  `overlap_and_add/2`, `fft_frequencies/2`, `mel_filters/4`, `stft_to_mel/3` — same names, option keys, defaults, return
  shapes and `ArgumentError`s as the reference (lib/nx_signal.ex:68-130, :154-166, :249-364, :397-513, :582-736) —
  plus `NxSignalAMD.Windows`, `NxSignalAMD.Filters`, `NxSignalAMD.Convolution`, device-resident tensors
  (`NxSignalAMD.DeviceTensor`) and multi-GPU sharding (`NxSignalAMD.Sharded`).

  The functions are ordinary `def`s over a dirty NIF (they cannot be traced inside someone else's `defn`); tensors cross
  as `Nx.to_binary/1` payloads (f32 / c64, row-major, little-endian) or stay in HBM as `NxSignalAMD.DeviceTensor`s.
  Vectorized (multichannel) inputs keep their vectorized axes like the reference's (`lib/nx_signal.ex:358-363`).

  The build image has no BEAM, so these modules are not compiled there; the NIF they call IS compiled and executed by
  the test-suite against a stand-in term runtime (tests/test_nif_shim.py), with the term shapes used below.
  """

  alias NxSignalAMD.{DeviceTensor, NIF}

  @pad %{valid: 0, reflect: 1, same: 2}
  @scaling %{nil => 0, :spectrum => 1, :psd => 2}

  @doc "One context per GPU (cached in :persistent_term)."
  def context(device \\ 0) do
    key = {__MODULE__, :ctx, device}

    case :persistent_term.get(key, nil) do
      nil ->
        {:ok, ctx} = NIF.ctx_create(device) |> unwrap!()
        :persistent_term.put(key, ctx)
        ctx

      ctx ->
        ctx
    end
  end

  @doc """
  Which kernel families did the last compute call on this context launch?  A `"+"`-separated string such as `"stft.pair"`,
  `"istft.wave.deep+istft.edge_chunks"`, `"fir.pair+fir.pair.edge"` or `"stft.generic.pow2"` (`nxsig_ctx_last_dispatch`,
  include/nxsig.h).  Diagnostic: a shape that silently leaves the tuned kernels shows up here.
  """
  def last_dispatch(ctx \\ context()) do
    {:ok, families} = NIF.last_dispatch(ctx) |> unwrap!()
    families
  end

  @doc """
  See `NxSignal.stft/3`. Returns `{z, times, frequencies}` with `z :: c64[frames: M][frequencies: K]`.

  `data` may be an `Nx.Tensor` (vectorized axes = channels, re-applied to `z`) or a `NxSignalAMD.DeviceTensor`
  (then `z` is a `DeviceTensor` too and nothing leaves the GPU).
  """
  def stft(data, window, opts \\ [])

  # f32 samples, or c64 samples (IQ data): the reference frames, multiplies and transforms whatever tensor it is given
  # (lib/nx_signal.ex:94-102); complex samples take one transform per frame (nxsig_stft_c64)
  def stft(%DeviceTensor{type: type} = data, window, opts) when type in [{:f, 32}, {:c, 64}] do
    {params, fft_length} = stft_params!(window, opts)
    {batch_shape, length} = split_last(data.shape)
    batch = Tuple.product(batch_shape)
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()

    {:ok, zref, m} =
      if type == {:c, 64} do
        NIF.stft_c64_dev(data.ctx, data.ref, length, batch, w, params)
      else
        NIF.stft_dev(data.ctx, data.ref, length, batch, w, params)
      end
      |> unwrap!()

    {t, f} = times_and_frequencies(params, m)

    z = %DeviceTensor{
      ref: zref,
      ctx: data.ctx,
      shape: append(batch_shape, [m, fft_length]),
      type: {:c, 64},
      names: List.duplicate(nil, tuple_size(batch_shape)) ++ [:frames, :frequencies]
    }

    {z, t, f}
  end

  def stft(%Nx.Tensor{} = data, window, opts) do
    {params, fft_length} = stft_params!(window, opts)
    {flat, vec_axes} = devectorize(data)
    {batch_shape, length} = split_last(Nx.shape(flat))
    batch = Tuple.product(batch_shape)
    # f64 samples or an f64 window: Nx.multiply promotes and Nx.fft returns c128 (lib/nx_signal.ex:101-102) -> the f64 tier
    wide = Nx.type(data) == {:f, 64} or Nx.type(window) == {:f, 64}

    complex = Nx.type(data) == {:c, 64}

    {:ok, z, m, t, f} =
      cond do
        complex and Nx.type(window) != {:f, 64} ->
          # c64 samples (IQ data): one transform per frame, c64 x f32 componentwise (lib/nx_signal.ex:101-102)
          NIF.stft_c64(context(), Nx.to_binary(flat), length, batch, window |> Nx.as_type(:f32) |> Nx.to_binary(), params)

        Nx.type(data) in [{:c, 64}, {:c, 128}] ->
          # c128 samples, or c64 samples under an f64 window (the product of :101 promotes): the f64 tier, complex x real componentwise
          {w, w64} = window_binary(window)
          NIF.stft_c128(context(), flat |> Nx.as_type(:c128) |> Nx.to_binary(), length, batch, w, w64, params)

        wide ->
          {w, w64} = window_binary(window)
          NIF.stft_f64(context(), flat |> Nx.as_type(:f64) |> Nx.to_binary(), length, batch, w, w64, params)

        true ->
          x = flat |> Nx.as_type(:f32) |> Nx.to_binary()
          NIF.stft(context(), x, length, batch, window |> Nx.as_type(:f32) |> Nx.to_binary(), params)
      end
      |> unwrap!()

    names = List.duplicate(nil, tuple_size(batch_shape)) ++ [:frames, :frequencies]

    z =
      Nx.from_binary(z, if(wide or Nx.type(data) == {:c, 128} or (complex and Nx.type(window) == {:f, 64}), do: :c128, else: :c64))
      |> Nx.reshape(append(batch_shape, [m, fft_length]), names: names)
      |> revectorize(vec_axes)

    {z, Nx.from_binary(t, :f32) |> Nx.reshape({m}, names: [:frames]),
     Nx.from_binary(f, :f32) |> Nx.reshape({fft_length}, names: [:frequencies])}
  end

  @doc """
  Extension (not in the reference API): `stft/3` on a device tensor restricted to the bins `0 .. fft_length/2 - 1` — the slice
  the reference's own downstream code keeps for real signals (`stft_to_mel/3`, the spectrogram guide) — written straight from
  the transform: half the output traffic, the same bits as the first half of `stft/3`'s rows.
  """
  def stft_onesided(%DeviceTensor{type: {:f, 32}} = data, window, opts) do
    {params, fft_length} = stft_params!(window, opts)
    {batch_shape, length} = split_last(data.shape)
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()

    {:ok, zref, m} =
      NIF.stft_onesided_dev(data.ctx, data.ref, length, Tuple.product(batch_shape), w, params) |> unwrap!()

    {t, f} = times_and_frequencies(params, m)
    half = div(fft_length, 2)

    z = %DeviceTensor{
      ref: zref,
      ctx: data.ctx,
      shape: append(batch_shape, [m, half]),
      type: {:c, 64},
      names: List.duplicate(nil, tuple_size(batch_shape)) ++ [:frames, :frequencies]
    }

    {z, t, Nx.slice(f, [0], [half])}
  end

  @doc """
  Extension (not in the reference API): the packed one-sided pair.  `stft_packed/3` is `stft_onesided/3` with nothing of a real
  frame's spectrum lost — the imaginary part of bin 0 (zero for a real frame) carries `Re X[fft_length / 2]`, the Nyquist bin —
  and `istft_packed/3` inverts exactly that layout into a REAL f32 signal: the reference's STFT-domain filtering chain
  (`guides/filtering.livemd:137-159`) at 4 KB + 1 KB of HBM traffic per 1024-point frame instead of 8 + 2.  A pointwise product
  of two packed tensors must treat bin 0 as the two reals it is (DC and Nyquist).
  """
  def stft_packed(%DeviceTensor{type: {:f, 32}} = data, window, opts) do
    {params, fft_length} = stft_params!(window, opts)
    {batch_shape, length} = split_last(data.shape)
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()

    {:ok, zref, m} =
      NIF.stft_packed_dev(data.ctx, data.ref, length, Tuple.product(batch_shape), w, params) |> unwrap!()

    {t, f} = times_and_frequencies(params, m)
    half = div(fft_length, 2)

    z = %DeviceTensor{
      ref: zref,
      ctx: data.ctx,
      shape: append(batch_shape, [m, half]),
      type: {:c, 64},
      names: List.duplicate(nil, tuple_size(batch_shape)) ++ [:frames, :frequencies]
    }

    {z, t, Nx.slice(f, [0], [half])}
  end

  @doc "Inverse of `stft_packed/3`: packed `c64[..., frames, fft_length / 2]` on the device -> real `f32[..., M * hop + overlap]`."
  def istft_packed(%DeviceTensor{type: {:c, 64}} = z, window, opts) do
    rank = tuple_size(z.shape)
    full_shape = put_elem(z.shape, rank - 1, 2 * elem(z.shape, rank - 1))
    {params, _overlap, m, batch_shape} = istft_params!(full_shape, window, opts)
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()
    {:ok, yref} = NIF.istft_packed_dev(z.ctx, z.ref, m, Tuple.product(batch_shape), w, params) |> unwrap!()
    out_len = m * elem(params, 1) + (elem(params, 0) - elem(params, 1))
    %DeviceTensor{ref: yref, ctx: z.ctx, shape: append(batch_shape, [out_len]), type: {:f, 32}, names: nil}
  end

  @doc "See `NxSignal.istft/3`. Returns a c64 tensor of length `M * hop + overlap_length` (complex, like the reference)."
  def istft(data, window, opts)

  def istft(%DeviceTensor{type: {:c, 64}} = data, window, opts) do
    {params, _overlap, m, batch_shape} = istft_params!(data.shape, window, opts)
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()
    {:ok, yref} = NIF.istft_dev(data.ctx, data.ref, m, Tuple.product(batch_shape), w, params) |> unwrap!()
    out_len = m * elem(params, 1) + (elem(params, 0) - elem(params, 1))
    %DeviceTensor{ref: yref, ctx: data.ctx, shape: append(batch_shape, [out_len]), type: {:c, 64}}
  end

  def istft(%Nx.Tensor{} = data, window, opts) do
    {flat, vec_axes} = devectorize(data)
    {params, _overlap, m, batch_shape} = istft_params!(Nx.shape(flat), window, opts)
    out_len = m * elem(params, 1) + (elem(params, 0) - elem(params, 1))

    cond do
      Nx.type(data) in [{:c, 128}, {:f, 64}] ->
        # a c128 spectrum is inverted in c128 (Nx.ifft, lib/nx_signal.ex:609): the f64 tier
        {w, w64} = window_binary(window)
        z = flat |> Nx.as_type(:c128) |> Nx.to_binary()
        {:ok, y} = NIF.istft_c128(context(), z, m, Tuple.product(batch_shape), w, w64, params) |> unwrap!()
        Nx.from_binary(y, :c128) |> Nx.reshape(append(batch_shape, [out_len])) |> revectorize(vec_axes)

      Nx.type(window) == {:f, 64} ->
        # the reference would invert in c64 and only then promote the frames: that mix is not modelled
        raise ArgumentError, "istft: a c64 spectrum with an f64 window is not built; pass the spectrum as c128"

      true ->
        z = flat |> Nx.as_type(:c64) |> Nx.to_binary()
        w = window |> Nx.as_type(:f32) |> Nx.to_binary()
        {:ok, y} = NIF.istft(context(), z, m, Tuple.product(batch_shape), w, params) |> unwrap!()
        Nx.from_binary(y, :c64) |> Nx.reshape(append(batch_shape, [out_len])) |> revectorize(vec_axes)
    end
  end

  # the window as the binary the f64 tier takes: f64 when the tensor is, else f32 (the :scaling scalar and the |w|^2
  # normaliser are formed in the window's own type, include/nxsig.h)
  defp window_binary(window) do
    if Nx.type(window) == {:f, 64},
      do: {Nx.to_binary(window), 1},
      else: {window |> Nx.as_type(:f32) |> Nx.to_binary(), 0}
  end

  @doc """
  `istft(Nx.multiply(z, h), window, opts)` in one library call: the last two steps of the reference's STFT-domain filtering
  workflow (`guides/filtering.livemd:141` and `:150-157`). `h :: c64[fft_length]` is the DFT of the filter. Bit-identical to
  the two calls; for 1024-point frames the product is formed inside the inverse-STFT kernel, so the filtered spectrogram is
  never written to HBM (2.4x faster than multiply-then-istft on 16 x 60 s of audio). `z` is not modified.
  """
  def istft_filtered(data, h, window, opts)

  def istft_filtered(%DeviceTensor{type: {:c, 64}} = data, %Nx.Tensor{} = h, window, opts) do
    {params, _overlap, m, batch_shape} = istft_params!(data.shape, window, opts)
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()
    hb = filter_spectrum!(h, elem(params, 2))

    {:ok, yref} =
      NIF.istft_filtered_dev(data.ctx, data.ref, m, Tuple.product(batch_shape), w, params, hb) |> unwrap!()

    out_len = m * elem(params, 1) + (elem(params, 0) - elem(params, 1))
    %DeviceTensor{ref: yref, ctx: data.ctx, shape: append(batch_shape, [out_len]), type: {:c, 64}}
  end

  def istft_filtered(%Nx.Tensor{} = data, %Nx.Tensor{} = h, window, opts) do
    {flat, vec_axes} = devectorize(data)
    {params, _overlap, m, batch_shape} = istft_params!(Nx.shape(flat), window, opts)
    z = flat |> Nx.as_type(:c64) |> Nx.to_binary()
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()
    hb = filter_spectrum!(h, elem(params, 2))
    {:ok, y} = NIF.istft_filtered(context(), z, m, Tuple.product(batch_shape), w, params, hb) |> unwrap!()
    out_len = m * elem(params, 1) + (elem(params, 0) - elem(params, 1))
    Nx.from_binary(y, :c64) |> Nx.reshape(append(batch_shape, [out_len])) |> revectorize(vec_axes)
  end

  defp filter_spectrum!(h, k) do
    if Nx.shape(h) != {k} do
      raise ArgumentError, "expected a filter spectrum of shape {#{k}}, got: #{inspect(Nx.shape(h))}"
    end

    h |> Nx.as_type(:c64) |> Nx.to_binary()
  end

  @doc "See `NxSignal.as_windowed/2` (lib/nx_signal.ex:249-364): `{..., L}` -> `{..., M, window_length}`, bit-exact gather."
  def as_windowed(%Nx.Tensor{} = tensor, opts \\ []) do
    opts = Keyword.validate!(opts, [:window_length, padding: :valid, stride: 1])
    window_length = opts[:window_length] || raise ArgumentError, "missing :window_length option"

    stride =
      case opts[:stride] do
        [s] when is_integer(s) and s >= 1 -> s
        s when is_integer(s) and s >= 1 -> s
        s -> raise ArgumentError, "expected an integer >= 1 or a list of integers, got: #{inspect(s)}"
      end

    {pad_mode, lo, hi} = padding!(opts[:padding])
    {flat, vec_axes} = devectorize(tensor)
    {batch_shape, length} = split_last(Nx.shape(flat))

    if Nx.type(tensor) == {:f, 64} do
      as_windowed_f64(flat, vec_axes, batch_shape, length, window_length, stride, pad_mode, lo, hi)
    else
      as_windowed_words32(tensor, flat, vec_axes, batch_shape, length, window_length, stride, pad_mode, lo, hi)
    end
  end

  defp as_windowed_f64(flat, vec_axes, batch_shape, length, window_length, stride, pad_mode, lo, hi) do
    {:ok, frames, m} =
      NIF.as_windowed_f64(context(), Nx.to_binary(flat), length, Tuple.product(batch_shape), window_length, stride, pad_mode, lo, hi)
      |> unwrap!()

    Nx.from_binary(frames, :f64) |> Nx.reshape(append(batch_shape, [m, window_length])) |> revectorize(vec_axes)
  end

  defp as_windowed_words32(tensor, flat, vec_axes, batch_shape, length, window_length, stride, pad_mode, lo, hi) do
    # a pure gather: 32-bit words travel untouched (bit-exact for f32 / s32 / u32); other integer types go through s32 when
    # every value fits, so that no integer is ever rounded through f32
    {words, word_type} = to_words32(flat)

    {:ok, frames, m} =
      NIF.as_windowed(context(), Nx.to_binary(words), length, Tuple.product(batch_shape), window_length, stride, pad_mode, lo, hi)
      |> unwrap!()

    Nx.from_binary(frames, word_type)
    |> Nx.reshape(append(batch_shape, [m, window_length]))
    |> Nx.as_type(Nx.type(tensor))
    |> revectorize(vec_axes)
  end

  defp to_words32(t) do
    case Nx.type(t) do
      {:f, 32} -> {t, {:f, 32}}
      {:s, 32} -> {t, {:s, 32}}
      {:u, 32} -> {t, {:u, 32}}
      {:f, 16} -> {Nx.as_type(t, :f32), {:f, 32}}
      {:bf, 16} -> {Nx.as_type(t, :f32), {:f, 32}}
      {kind, _} when kind in [:s, :u] ->
        lo = t |> Nx.reduce_min() |> Nx.to_number()
        hi = t |> Nx.reduce_max() |> Nx.to_number()

        if lo < -2_147_483_648 or hi > 2_147_483_647 do
          raise ArgumentError, "as_windowed: integer values beyond 32 bits are not supported by the MI355X path"
        end

        {Nx.as_type(t, :s32), {:s, 32}}

      other ->
        raise ArgumentError, "as_windowed: unsupported tensor type #{inspect(other)}"
    end
  end

  @doc "See `NxSignal.overlap_and_add/2` (lib/nx_signal.ex:684-736): `{..., M, N}` -> `{..., M * hop + overlap_length}`."
  def overlap_and_add(%Nx.Tensor{} = tensor, opts \\ []) do
    opts = Keyword.validate!(opts, [:overlap_length, type: Nx.type(tensor)])
    overlap_length = opts[:overlap_length] || raise ArgumentError, "missing :overlap_length option"
    {flat, vec_axes} = devectorize(tensor)
    shape = Nx.shape(flat)
    rank = tuple_size(shape)
    {m, n} = {elem(shape, rank - 2), elem(shape, rank - 1)}
    batch_shape = shape |> Tuple.delete_at(rank - 1) |> Tuple.delete_at(rank - 2)

    # f64 / c128 tensors are added in double by the f64 tier; everything else through f32 / c64
    {bin_type, components, nif} =
      case {Nx.type(tensor), Nx.Type.normalize!(opts[:type])} do
        {{:c, 128}, _} -> {:c128, 2, &NIF.overlap_and_add_f64/7}
        {{:f, 64}, _} -> {:f64, 1, &NIF.overlap_and_add_f64/7}
        {_, {:c, _}} -> {:c64, 2, &NIF.overlap_and_add/7}
        _ -> {:f32, 1, &NIF.overlap_and_add/7}
      end

    frames = flat |> Nx.as_type(bin_type) |> Nx.to_binary()
    {:ok, out} = nif.(context(), frames, m, Tuple.product(batch_shape), n, overlap_length, components) |> unwrap!()

    out_len = m * (n - overlap_length) + overlap_length

    Nx.from_binary(out, bin_type)
    |> Nx.reshape(append(batch_shape, [out_len]))
    |> Nx.as_type(opts[:type])
    |> revectorize(vec_axes)
  end

  @doc "See `NxSignal.fft_frequencies/2` (lib/nx_signal.ex:154-166)."
  def fft_frequencies(sampling_rate, opts \\ []) do
    opts = Keyword.validate!(opts, [:fft_length, :name, type: {:f, 32}, endpoint: false])
    fft_length = opts[:fft_length] || raise ArgumentError, "missing :fft_length option"
    endpoint = if(opts[:endpoint], do: 1, else: 0)

    if Nx.Type.normalize!(opts[:type]) == {:f, 64} do
      {:ok, bin} = NIF.fft_frequencies_f64(sampling_rate * 1.0, fft_length, endpoint) |> unwrap!()
      Nx.from_binary(bin, :f64) |> Nx.reshape({fft_length}, names: [opts[:name]])
    else
      {:ok, bin} = NIF.fft_frequencies(sampling_rate * 1.0, fft_length, endpoint) |> unwrap!()
      Nx.from_binary(bin, :f32) |> Nx.reshape({fft_length}, names: [opts[:name]]) |> Nx.as_type(opts[:type])
    end
  end

  @doc "See `NxSignal.mel_filters/4` (lib/nx_signal.ex:397-445): `f32[mels: mel_bins][frequencies: fft_length]`."
  def mel_filters(fft_length, mel_bins, sampling_rate, opts \\ []) do
    opts = Keyword.validate!(opts, max_mel: 3016, mel_frequency_spacing: 200 / 3, type: {:f, 32})

    {:ok, bin} =
      NIF.mel_filters(fft_length, mel_bins, sampling_rate * 1.0, opts[:max_mel] * 1.0, opts[:mel_frequency_spacing] * 1.0)
      |> unwrap!()

    Nx.from_binary(bin, :f32) |> Nx.reshape({mel_bins, fft_length}, names: [:mels, :frequencies]) |> Nx.as_type(opts[:type])
  end

  @doc "See `NxSignal.stft_to_mel/3` (lib/nx_signal.ex:486-513): `c64[frames][frequencies]` -> `f32[frames][mel]`."
  def stft_to_mel(%Nx.Tensor{} = z, sampling_rate, opts \\ []) do
    opts = Keyword.validate!(opts, [:fft_length, :mel_bins, :max_mel, :mel_frequency_spacing, type: {:f, 32}])
    {flat, vec_axes} = devectorize(z)
    shape = Nx.shape(flat)
    rank = tuple_size(shape)
    {m, k} = {elem(shape, rank - 2), elem(shape, rank - 1)}
    fft_length = opts[:fft_length] || k
    mel_bins = opts[:mel_bins] || raise ArgumentError, "missing :mel_bins option"
    batch_shape = shape |> Tuple.delete_at(rank - 1) |> Tuple.delete_at(rank - 2)
    mel_opts = Keyword.take(opts, [:max_mel, :mel_frequency_spacing]) |> Keyword.reject(fn {_, v} -> is_nil(v) end)
    filters = mel_filters(fft_length, mel_bins, sampling_rate, mel_opts) |> Nx.to_binary()
    rows = Tuple.product(batch_shape) * m
    zb = flat |> Nx.as_type(:c64) |> Nx.to_binary()
    {:ok, out} = NIF.stft_to_mel(context(), zb, rows, fft_length, mel_bins, filters) |> unwrap!()

    Nx.from_binary(out, :f32)
    |> Nx.reshape(append(batch_shape, [m, mel_bins]), names: List.duplicate(nil, tuple_size(batch_shape)) ++ [:frames, :mel])
    |> Nx.as_type(opts[:type])
    |> revectorize(vec_axes)
  end

  @doc """
  Fused `stft/3 |> stft_to_mel/3`: the log-mel spectrogram without writing the complex spectrum to HBM
  (`mel_bins * 4` bytes per frame leave the chip instead of `fft_length * 8`). Returns `f32[..., frames, mel]`.
  """
  def mel_spectrogram(%Nx.Tensor{} = data, window, opts \\ []) do
    {mel_opts, stft_opts} = Keyword.split(opts, [:mel_bins, :max_mel, :mel_frequency_spacing])
    mel_bins = mel_opts[:mel_bins] || raise ArgumentError, "missing :mel_bins option"
    {params, fft_length} = stft_params!(window, stft_opts)
    filters = mel_filters(fft_length, mel_bins, elem(params, 7), Keyword.delete(mel_opts, :mel_bins)) |> Nx.to_binary()
    {flat, vec_axes} = devectorize(data)
    {batch_shape, length} = split_last(Nx.shape(flat))
    x = flat |> Nx.as_type(:f32) |> Nx.to_binary()
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()

    {:ok, out, m} =
      NIF.stft_mel(context(), x, length, Tuple.product(batch_shape), w, params, mel_bins, filters) |> unwrap!()

    Nx.from_binary(out, :f32) |> Nx.reshape(append(batch_shape, [m, mel_bins])) |> revectorize(vec_axes)
  end

  @doc """
  Magnitude spectrogram fused with the STFT (what `guides/spectrogram.livemd:76-92` derives from `NxSignal.stft/3`): `Nx.abs(s)`
  of the bins below `fft_length / 2` (`kind: :magnitude`), `|s|^2` (`:power`) or `20 * log10(|s| / max |s|)` (`:dbfs`), without
  writing the complex spectrum to HBM. Returns `f32[..., frames, fft_length / 2]`. Opt-in: not a reference function.
  """
  def spectrogram(%Nx.Tensor{} = data, window, opts \\ []) do
    {kind, stft_opts} = Keyword.pop(opts, :kind, :magnitude)

    kind_code =
      case kind do
        :magnitude -> 0
        :power -> 1
        :dbfs -> 2
        other -> raise ArgumentError, "expected :kind to be one of :magnitude, :power, :dbfs, got: #{inspect(other)}"
      end

    {params, fft_length} = stft_params!(window, stft_opts)
    {flat, vec_axes} = devectorize(data)
    {batch_shape, length} = split_last(Nx.shape(flat))
    x = flat |> Nx.as_type(:f32) |> Nx.to_binary()
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()

    {:ok, out, m} =
      NIF.stft_magnitude(context(), x, length, Tuple.product(batch_shape), w, params, kind_code) |> unwrap!()

    Nx.from_binary(out, :f32) |> Nx.reshape(append(batch_shape, [m, div(fft_length, 2)])) |> revectorize(vec_axes)
  end

  # ------------------------------------------------------------------------------------------ shared helpers
  @doc false
  def unwrap!({:error, {-1, msg}}), do: raise(ArgumentError, msg)
  def unwrap!({:error, {code, msg}}), do: raise(RuntimeError, "nxsig status #{code}: #{msg}")
  def unwrap!(ok), do: ok

  @doc false
  # option parsing / defaults of NxSignal.stft/3 (lib/nx_signal.ex:71-85; quirks B1-B3 of SURVEY App. B)
  def stft_params!(window, opts) do
    {frame_length} = Nx.shape(window)

    opts =
      Keyword.validate!(opts, [
        :overlap_length,
        :window,
        :scaling,
        window_padding: :valid,
        sampling_rate: 100,
        fft_length: :power_of_two
      ])

    sampling_rate = opts[:sampling_rate] || raise ArgumentError, "missing sampling_rate option"
    overlap_length = opts[:overlap_length] || div(frame_length, 2)
    fft_length = resolve_fft_length(opts[:fft_length], frame_length)
    {pad_mode, lo, hi} = padding!(opts[:window_padding])
    scaling = scaling!(opts[:scaling])
    {{frame_length, frame_length - overlap_length, fft_length, pad_mode, lo, hi, scaling, sampling_rate * 1.0}, fft_length}
  end

  @doc false
  def istft_params!(shape, window, opts) do
    opts = Keyword.validate!(opts, [:fft_length, :overlap_length, :scaling, sampling_rate: 1000])
    {frame_length} = Nx.shape(window)
    overlap_length = opts[:overlap_length] || div(frame_length, 2)
    scaling = scaling!(opts[:scaling])

    if opts[:scaling] == :psd and is_nil(opts[:sampling_rate]) do
      raise ArgumentError, ":sampling_rate is mandatory if scaling is :psd"
    end

    if overlap_length >= frame_length do
      raise ArgumentError,
            "overlap_length must be a number less than the window size #{frame_length}, got: #{inspect(frame_length)}"
    end

    rank = tuple_size(shape)
    {m, k} = {elem(shape, rank - 2), elem(shape, rank - 1)}
    batch_shape = shape |> Tuple.delete_at(rank - 1) |> Tuple.delete_at(rank - 2)
    fft_length = resolve_fft_length(opts[:fft_length] || :power_of_two, k)
    params = {frame_length, frame_length - overlap_length, fft_length, 0, 0, 0, scaling, (opts[:sampling_rate] || 0) * 1.0}
    {params, overlap_length, m, batch_shape}
  end

  @doc false
  def times_and_frequencies({frame_length, _hop, fft_length, _pad, _lo, _hi, _scaling, fs}, m) do
    # times = linspace(N / (2 fs), M N / (2 fs), n: M) (lib/nx_signal.ex:108-111, quirk B4), f32 arithmetic like the reference
    step = frame_length / (2 * fs)
    times = Nx.linspace(step, step * m, n: m, name: :frames, type: :f32)
    {times, fft_frequencies(fs, fft_length: fft_length, name: :frequencies)}
  end

  # vectorized axes = channels: flatten them into leading axes for the NIF, re-apply them to the result
  defp devectorize(%Nx.Tensor{vectorized_axes: []} = t), do: {t, []}
  defp devectorize(%Nx.Tensor{vectorized_axes: axes} = t), do: {Nx.devectorize(t, keep_names: false), axes}
  defp revectorize(t, []), do: t
  defp revectorize(t, axes), do: Nx.vectorize(t, axes)

  defp append(shape, dims), do: Enum.reduce(dims, shape, fn d, acc -> Tuple.insert_at(acc, tuple_size(acc), d) end)

  defp resolve_fft_length(:power_of_two, n), do: next_pow2(n, 1)
  defp resolve_fft_length(k, _n) when is_integer(k) and k >= 1, do: k

  defp resolve_fft_length(other, _n),
    do: raise(ArgumentError, "expected :fft_length to be a positive integer or :power_of_two, got: #{inspect(other)}")

  defp next_pow2(n, p) when p >= n, do: p
  defp next_pow2(n, p), do: next_pow2(n, p * 2)

  @doc false
  def padding!(mode) when is_map_key(@pad, mode), do: {@pad[mode], 0, 0}
  def padding!([{lo, hi}]) when is_integer(lo) and is_integer(hi), do: {3, lo, hi}

  def padding!(mode) when is_list(mode),
    do: raise(ArgumentError, "padding must be a list of {high, low} tuples, where each element is an integer. Got: #{inspect(mode)}")

  def padding!(mode),
    do:
      raise(
        ArgumentError,
        "invalid padding mode specified, padding must be one of :valid, :same, or a padding configuration, got: #{inspect(mode)}"
      )

  defp scaling!(s) when is_map_key(@scaling, s), do: @scaling[s]

  defp scaling!(s),
    do: raise(ArgumentError, "invalid :scaling, expected one of :spectrum, :psd or nil, got: #{inspect(s)}")

  @doc false
  def split_last(shape) do
    r = tuple_size(shape)
    {Tuple.delete_at(shape, r - 1), elem(shape, r - 1)}
  end
end
