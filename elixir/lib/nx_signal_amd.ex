defmodule NxSignalAMD do
  @moduledoc """
  Drop-in for the hot path of `NxSignal` on AMD Instinct MI355X: `stft/3`, `istft/3` (same option names,
  defaults, return shapes and `ArgumentError`s as `NxSignal` v0.3.0 — lib/nx_signal.ex:68-130, :582-638),
  plus `NxSignalAMD.Windows`, `NxSignalAMD.Filters.firwin/3` and the new `NxSignalAMD.Filters.fir/3`.

  The functions are ordinary `def`s over a dirty NIF (they cannot be traced inside someone else's `defn`);
  tensors cross as `Nx.to_binary/1` payloads (f32 / c64, row-major, little-endian) or stay in HBM as
  `NxSignalAMD.DeviceTensor` resources.  NOT compiled in the build image (no BEAM there) — INTEGRATION.md.
  """

  alias NxSignalAMD.NIF

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

  @doc "See `NxSignal.stft/3`. Returns `{z, times, frequencies}` with `z :: c64[frames: M][frequencies: K]`."
  def stft(data, window, opts \\ []) do
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
    {batch_shape, length} = split_last(Nx.shape(data))
    batch = Tuple.product(batch_shape)

    params =
      {frame_length, frame_length - overlap_length, fft_length, pad_mode, lo, hi, scaling, sampling_rate * 1.0}

    x = data |> Nx.as_type(:f32) |> Nx.to_binary()
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()

    {:ok, z, m, t, f} = NIF.stft(context(), x, length, batch, w, params) |> unwrap!()

    z_shape = batch_shape |> Tuple.insert_at(tuple_size(batch_shape), m) |> Tuple.insert_at(tuple_size(batch_shape) + 1, fft_length)
    names = List.duplicate(nil, tuple_size(batch_shape)) ++ [:frames, :frequencies]

    {Nx.from_binary(z, :c64) |> Nx.reshape(z_shape, names: names),
     Nx.from_binary(t, :f32) |> Nx.reshape({m}, names: [:frames]),
     Nx.from_binary(f, :f32) |> Nx.reshape({fft_length}, names: [:frequencies])}
  end

  @doc "See `NxSignal.istft/3`. Returns a c64 tensor of length `M * hop + overlap_length`."
  def istft(data, window, opts) do
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

    shape = Nx.shape(data)
    rank = tuple_size(shape)
    {m, k} = {elem(shape, rank - 2), elem(shape, rank - 1)}
    batch_shape = shape |> Tuple.delete_at(rank - 1) |> Tuple.delete_at(rank - 2)
    batch = Tuple.product(batch_shape)
    fft_length = resolve_fft_length(opts[:fft_length] || :power_of_two, k)
    params = {frame_length, frame_length - overlap_length, fft_length, 0, 0, 0, scaling, (opts[:sampling_rate] || 0) * 1.0}
    z = data |> Nx.as_type(:c64) |> Nx.to_binary()
    w = window |> Nx.as_type(:f32) |> Nx.to_binary()
    {:ok, y} = NIF.istft(context(), z, m, batch, w, params) |> unwrap!()
    out_len = m * (frame_length - overlap_length) + overlap_length
    Nx.from_binary(y, :c64) |> Nx.reshape(Tuple.insert_at(batch_shape, tuple_size(batch_shape), out_len))
  end

  @doc false
  def unwrap!({:error, {-1, msg}}), do: raise(ArgumentError, msg)
  def unwrap!({:error, {code, msg}}), do: raise(RuntimeError, "nxsig status #{code}: #{msg}")
  def unwrap!(ok), do: ok

  defp resolve_fft_length(:power_of_two, n), do: next_pow2(n, 1)
  defp resolve_fft_length(k, _n) when is_integer(k) and k >= 1, do: k

  defp resolve_fft_length(other, _n),
    do: raise(ArgumentError, "expected :fft_length to be a positive integer or :power_of_two, got: #{inspect(other)}")

  defp next_pow2(n, p) when p >= n, do: p
  defp next_pow2(n, p), do: next_pow2(n, p * 2)

  defp padding!(mode) when is_map_key(@pad, mode), do: {@pad[mode], 0, 0}
  defp padding!([{lo, hi}]) when is_integer(lo) and is_integer(hi), do: {3, lo, hi}

  defp padding!(mode) when is_list(mode),
    do: raise(ArgumentError, "padding must be a list of {high, low} tuples, where each element is an integer. Got: #{inspect(mode)}")

  defp padding!(mode),
    do:
      raise(
        ArgumentError,
        "invalid padding mode specified, padding must be one of :valid, :same, or a padding configuration, got: #{inspect(mode)}"
      )

  defp scaling!(s) when is_map_key(@scaling, s), do: @scaling[s]

  defp scaling!(s),
    do: raise(ArgumentError, "invalid :scaling, expected one of :spectrum, :psd or nil, got: #{inspect(s)}")

  defp split_last(shape) do
    r = tuple_size(shape)
    {Tuple.delete_at(shape, r - 1), elem(shape, r - 1)}
  end
end
